"""Is the long-sequence path (several launches per call) launch/CPU-bound?  Eager vs HIP-graph replay of the forward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (N, B, H, L) in ((4194304, 1, 16, 1048576), (65536, 16, 768, 32768), (32768, 16, 768, 16384), (4096, 16, 768, 2048)):
    u = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda().eval()
    with torch.no_grad():
        te = ev(lambda: mod(u, k))
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): mod(u, k)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = mod(u, k)
        tg = ev(g.replay)
        ref = mod(u, k)
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(ref, y)
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(20): mod(u, k)
    cpu = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    print(f"fft={N} B={B} H={H} L={L}: eager {te:.4f} ms  graph replay {tg:.4f} ms  (CPU enqueue time per call {cpu:.4f} ms)  same={same}", flush=True)
