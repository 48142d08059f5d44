"""A/B of the streaming-row policy (FFC_STREAM unset = library policy, 0 = off, 1 = forced on), same process, same box."""
import os, sys, torch
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
for (N, B, H, L, gated) in ((16384, 8, 1024, 8192, True), (16384, 16, 768, 8192, False), (32768, 16, 768, 16384, False), (32768, 8, 768, 16384, True), (4096, 16, 768, 2048, False)):
    u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True); k = torch.randn(H, L, device="cuda").requires_grad_(True)
    g = [torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True) for _ in range(2)] if gated else []
    dout = torch.randn(B, H, L, device="cuda").bfloat16()
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()
    y = mod(u, k, *g)
    def bwd():
        for t in [u, k] + g: t.grad = None
        y.backward(dout, retain_graph=True)
    for rep in range(2):
        for mode in (None, "0", "1"):
            if mode is None: os.environ.pop("FFC_STREAM", None)
            else: os.environ["FFC_STREAM"] = mode
            with torch.no_grad():
                tf = ev(lambda: mod(u, k, *g))
            tb = ev(bwd)
            print(f"N={N} B={B} H={H} L={L} gated={gated} FFC_STREAM={mode}: fwd {tf:.4f} bwd {tb:.4f}", flush=True)
    os.environ.pop("FFC_STREAM", None); __import__("flashfftconv.conv").conv.reload_env()
