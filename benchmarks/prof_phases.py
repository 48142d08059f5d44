"""Per-phase cycle breakdown of the N=32768 forward kernel (profiling build, s_memtime)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
N, B, H, L = 32768, 16, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda")
u = torch.randn(B, H, L, device=dev).to(torch.bfloat16); k = torch.randn(H, L, device=dev)
mod = FlashFFTConv(N, dtype=torch.bfloat16).to(dev)
plan = mod._get_plan(dev); kf = C._kernel_fft(plan, k)
y = torch.empty_like(u)
lib = _lib.lib()
prof = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
grid = ctypes.c_int()
for _ in range(2):
    prof.zero_()
    _lib.check(lib.ffc_conv_fwd_prof(plan.handle, _lib.ptr(u), _lib.ptr(kf), _lib.ptr(y), B, H, L, _lib.ptr(prof), ctypes.byref(grid), None), "prof")
torch.cuda.synchronize()
p = prof[: grid.value * 64].view(grid.value, 8, 8).double()
tot = p.sum(-1)
names = ["rows_in", "phaseA", "barrier1", "phaseB", "barrier2", "phaseC", "rows_out", "-"]
jobs = (B + 1) // 2
print(f"grid={grid.value} FLAGS={os.environ.get('FFC_FLAGS','0')} cycles per wave per pair (mean over waves), total {tot.mean().item()/jobs:.0f}")
for i, n in enumerate(names[:7]):
    print(f"  {n:9s} {p[..., i].mean().item()/jobs:9.0f}  ({100*p[..., i].sum().item()/tot.sum().item():5.1f}%)   min {p[...,i].min().item()/jobs:8.0f} max {p[...,i].max().item()/jobs:8.0f}")
# per wave index (SIMD = wave % 4): is the barrier skew systematic?
print("per wave index, mean over workgroups (cycles per pair):")
print("  wave " + " ".join(f"{n:>9s}" for n in names[:7]))
for w in range(8):
    print(f"  {w:4d} " + " ".join(f"{p[:, w, i].mean().item()/jobs:9.0f}" for i in range(7)))
