"""Profiling driver: one shape, forward only or forward + backward through the module (run under rocprofv3 --stats).
usage: prof_one.py N B H L [fwd|both] [gated|plain] [float16|bfloat16]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv
N, B, H, L = (int(x) for x in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "both"
gated = len(sys.argv) > 6 and sys.argv[6] == "gated"
dt = getattr(torch, sys.argv[7]) if len(sys.argv) > 7 else torch.bfloat16
u = torch.randn(B, H, L, device="cuda").to(dt).requires_grad_(True); k = torch.randn(H, L, device="cuda").requires_grad_(True)
g = [torch.randn(B, H, L, device="cuda").to(dt).requires_grad_(True) for _ in range(2)] if gated else []
dout = torch.randn(B, H, L, device="cuda").to(dt)
mod = FlashFFTConv(N, dtype=dt).cuda()
for _ in range(6):
    if mode == "fwd":
        with torch.no_grad():
            mod(u, k, *g)
    else:
        for t in [u, k] + g: t.grad = None
        mod(u, k, *g).backward(dout)
torch.cuda.synchronize()
