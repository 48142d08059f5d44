"""Profiling driver: config-2 forward conv kernel only (k_f prepared once), a few launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C
N, B, H, L = 32768, 16, 768, 16384
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dev = torch.device("cuda")
u = torch.randn(B, H, L, device=dev).to(torch.bfloat16); k = torch.randn(H, L, device=dev)
mod = FlashFFTConv(N, dtype=torch.bfloat16).to(dev)
plan = mod._get_plan(dev); kf = C._kernel_fft(plan, k)
for _ in range(4):
    y = C._conv(plan, u, kf, None, None, False)
torch.cuda.synchronize()
