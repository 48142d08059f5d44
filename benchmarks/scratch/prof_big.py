"""Profiling driver: one big size (fft 65536 by default), forward + backward through the module."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
B, H = 16, int(sys.argv[2]) if len(sys.argv) > 2 else 768
L = N // 2
u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True); k = torch.randn(H, L, device="cuda").requires_grad_(True)
dout = torch.randn(B, H, L, device="cuda").bfloat16()
mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()
for _ in range(4):
    u.grad = None; k.grad = None
    mod(u, k).backward(dout)
torch.cuda.synchronize()
