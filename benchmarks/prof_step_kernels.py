"""Profiling driver (rocprofv3 --pmc / --kernel-trace): the two big kernels of the default config-2 training step, a few launches each.
argv[1]: fwd = spectrum-saving forward incl. k -> k_f (ffc_conv_fwd_k: conv_kernel<...,SZ>), bwd = fused backward on saved spectra incl.
the dk tail (ffc_conv_bwd_k: bwd_kernel<..., ZM = 1>)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
N, B, H, L = 32768, 16, 768, 16384
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
plan = FlashFFTConv(N, dtype=torch.bfloat16).cuda()._get_plan(u.device)
kf = C._kernel_fft(plan, k)
z = torch.empty(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
y = torch.empty_like(u); du = torch.empty_like(u)
dk = torch.empty(H, L, device="cuda")
# the module's two launches (round 4): forward incl. k -> k_f of the head, backward incl. the dk tail
_lib.check(lib.ffc_conv_fwd_k(plan.handle, P(k), L, P(kf), P(u), None, None, P(y), P(z), None, B, H, L, sp()), "fwd_k")
for _ in range(4):
    if mode == "fwd":
        _lib.check(lib.ffc_conv_fwd_k(plan.handle, P(k), L, P(kf), P(u), None, None, P(y), P(z), None, B, H, L, sp()), "fwd_k")
    else:
        _lib.check(lib.ffc_conv_bwd_k(plan.handle, P(dout), P(u), P(kf), None, None, P(du), None, None, P(ws), P(z), None, P(dk), L, B, H, L, sp()), "bwd_k")
torch.cuda.synchronize()
