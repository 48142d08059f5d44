"""Per-phase cycle breakdown of the fused backward kernel ON SAVED SPECTRA incl. the dk tail (ffc_conv_bwd_k: bwd_kernel<.., ZM = 1>), the
dominant kernel of the default training step.  Profiling variant: `benchmarks/mkvariant.sh bwdprof "ffc_k_bwd.hip ffc_k_bwdz.hip"
-DFFC_BWD_PROF`, run with FFC_LIB=<that .so>.  s_memtime sums per wave, config 2 by default (B16 H768 L16384, fft 32768, bf16)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
B, H = 16, 768
L = int(sys.argv[2]) if len(sys.argv) > 2 else N // 2
u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
plan = FlashFFTConv(N, dtype=torch.bfloat16).cuda()._get_plan(u.device)
kf = C._kernel_fft(plan, k)
z = torch.empty(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
y = torch.empty_like(u); du = torch.empty_like(u); dk = torch.empty(H, L, device="cuda")
_lib.check(lib.ffc_conv_fwd_k(plan.handle, P(k), L, P(kf), P(u), None, None, P(y), P(z), None, B, H, L, sp()), "fwd_k")
for _ in range(3):
    _lib.check(lib.ffc_conv_bwd_k(plan.handle, P(dout), P(u), P(kf), None, None, P(du), None, None, P(ws), P(z), None, P(dk), L, B, H, L, sp()), "bwd_k")
torch.cuda.synchronize()
nwg = ((H + 7) // 8 * 8) * int(lib.ffc_dkf_slab_count(plan.handle, B, H))
lib.ffc_debug_bwd_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64]
host = torch.zeros(nwg * 8 * 16, dtype=torch.int64)
_lib.check(lib.ffc_debug_bwd_prof(ctypes.c_void_p(host.data_ptr()), host.numel()), "prof")
p = host.view(nwg, 8, 16).double()
p = p[p.sum((1, 2)) > 0]
names = {5: "(loop head)", 6: "rows_in dout (DMA wait)", 7: "phaseA dout", 8: "barrier", 9: "phaseB (fft, dk_f, dx)", 10: "barrier", 11: "phaseC", 12: "rows_out du"}
pairs = (B + 1) // 2 / max(1, int(lib.ffc_dkf_slab_count(plan.handle, B, H)))
tot = p.sum(-1)
print(f"bwd on saved spectra, fft {N} L {L}: {p.shape[0]} workgroups, cycles per wave per pair (mean over waves), total {tot.mean().item() / pairs:.0f}")
for i, n in names.items():
    print(f"  {n:26s} {p[..., i].mean().item() / pairs:9.0f}  ({100 * p[..., i].sum().item() / tot.sum().item():5.1f}%)   min {p[..., i].min().item() / pairs:8.0f} max {p[..., i].max().item() / pairs:8.0f}")
print("per wave index, mean over workgroups (cycles per pair):")
print("  wave " + " ".join(f"{names[i][:10]:>10s}" for i in names))
for w in range(8):
    print(f"  {w:4d} " + " ".join(f"{p[:, w, i].mean().item() / pairs:10.0f}" for i in names))
