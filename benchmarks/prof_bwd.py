"""Profiling driver: the fused backward kernel of config 2 (B16 H768 L16384, fft 32768, bf16), 4 launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib()
N, B, H, L = 32768, 16, 768, 16384
u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
kf = C._kernel_fft(plan, k)
ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
du = torch.empty_like(u)
for _ in range(4):
    _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(du), None, _lib.ptr(ws), B, H, L, _lib.stream_ptr()), "bwd")
torch.cuda.synchronize()
