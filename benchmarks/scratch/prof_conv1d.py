"""Timing of FlashDepthWiseConv1d (BHL and BLH, a few dtypes / kernel sizes): ms and algorithmic TB/s."""
import os, sys, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashDepthWiseConv1d
def ev(fn, it=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
B, D, L = 64, 2048, 8192
for bhl in (True, False):
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        for K in (3, 7):
            ref = nn.Conv1d(D, D, K, groups=D, padding=K // 2).cuda()
            m = FlashDepthWiseConv1d(D, K, K // 2, ref.weight.detach(), ref.bias.detach(), is_bhl=bhl, device="cuda", dtype=dt)
            x = torch.randn((B, D, L) if bhl else (B, L, D), device="cuda", dtype=dt, requires_grad=True)
            with torch.no_grad():
                tf = ev(lambda: m(x))
            y = m(x); do = torch.randn_like(y)
            def bw():
                x.grad = None
                for p in m.parameters(): p.grad = None
                y.backward(do, retain_graph=True)
            tb = ev(bw, 5)
            byts = B * D * L * x.element_size()
            print(f"bhl={bhl} {str(dt)[6:]:9s} K={K}: fwd {tf:.3f} ms ({2*byts/tf/1e9:.2f} TB/s)  bwd {tb:.3f} ms ({3*byts/tb/1e9:.2f} TB/s)", flush=True)
            del x, y, do
