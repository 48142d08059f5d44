"""Per-phase cycle breakdown of the fused backward kernel (profiling variant: `python flash-fft-conv_amd/build.py --variant bwdprof
-DFFC_BWD_PROF`, run with FFC_LIB=<that .so>).  s_memtime sums per wave, config 2 by default (B16 H768 L16384, fft 32768, bf16)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
B, H = 16, 768
L = int(sys.argv[2]) if len(sys.argv) > 2 else N // 2
gated = len(sys.argv) > 3 and sys.argv[3] == "gated"
u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
pre = torch.randn_like(u) if gated else None; post = torch.randn_like(u) if gated else None
mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
kf = C._kernel_fft(plan, k)
ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
du = torch.empty_like(u); dpre = torch.empty_like(u) if gated else None; dpost = torch.empty_like(u) if gated else None
for _ in range(3):
    _lib.check(lib.ffc_conv_bwd_gated(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pre), _lib.ptr(post), _lib.ptr(du), _lib.ptr(dpre),
                                      _lib.ptr(dpost), _lib.ptr(ws), B, H, L, _lib.stream_ptr()), "bwd")
torch.cuda.synchronize()
nwg = ((H + 7) // 8 * 8) * int(lib.ffc_dkf_slab_count(plan.handle, B, H))
lib.ffc_debug_bwd_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64]
host = torch.zeros(nwg * 8 * 16, dtype=torch.int64)
_lib.check(lib.ffc_debug_bwd_prof(ctypes.c_void_p(host.data_ptr()), host.numel()), "prof")
p = host.view(nwg, 8, 16).double()
p = p[p.sum((1, 2)) > 0]
names = ["rows_in u", "phaseA u", "barrier", "phaseB1 (fft u, scratch)", "barrier", "dpost part", "rows_in dout", "phaseA dout", "barrier",
         "phaseB2 (fft, dk_f, dx)", "barrier", "phaseC", "rows_out"]
pairs = (B + 1) // 2 / max(1, int(lib.ffc_dkf_slab_count(plan.handle, B, H)))
tot = p.sum(-1)
print(f"fft {N} L {L} gated={gated}: {p.shape[0]} workgroups, cycles per wave per pair (mean over waves), total {tot.mean().item() / pairs:.0f}")
for i, n in enumerate(names):
    print(f"  {n:26s} {p[..., i].mean().item() / pairs:9.0f}  ({100 * p[..., i].sum().item() / tot.sum().item():5.1f}%)   min {p[..., i].min().item() / pairs:8.0f} max {p[..., i].max().item() / pairs:8.0f}")
