"""Root-cause probe for the v_sin / v_cos hazard that DevB::settle() guards (csrc/ffc_dev.h): run the kernels of a library
variant many times on identical inputs and count launches whose results differ bitwise from the first one, and compare with
the fp64 oracle.  Variants (flash-fft-conv_amd/build.py --variant ...):
   base            settle() = s_nop 4 after v_sin/v_cos (product)
   nosettle        -DFFC_NO_SETTLE              packed fp32 consumers right behind the transcendentals
   nosettle_nopk   -DFFC_NO_SETTLE -DFFC_NO_PK  scalar fp32 consumers instead
usage: FFC_LIB=<variant .so> python benchmarks/hazard_probe.py"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import conv as C, _lib
from oracle.torch_ref import ref_fft_conv
lib = _lib.lib()
res = {}
for (N, B, H) in ((4096, 16, 64), (8192, 8, 64), (16384, 8, 64), (32768, 8, 64), (65536, 4, 32)):
    torch.manual_seed(0)
    L = N // 2
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16()
    k = torch.randn(H, L, device="cuda") * 0.1
    plan = C.get_plan(N, torch.bfloat16, u.device)
    kf0 = C._kernel_fft(plan, k).clone()
    y0 = C._conv(plan, u, kf0, None, None, False).clone()
    nb = lib.ffc_dkf_workspace_bytes(plan.handle, B, H)
    def dk_once():
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda"); du = torch.empty_like(u)
        _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf0), None, None, _lib.ptr(du), None, _lib.ptr(ws), B, H, L, None), "bwd")
        dk = torch.zeros(H, L, dtype=torch.float32, device="cuda")
        _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, L, _lib.ptr(dk), None), "dk")
        return du, dk
    du0, dk0 = dk_once()
    bad = {"kfft": 0, "conv": 0, "du": 0, "dk": 0}
    for _ in range(40):
        bad["kfft"] += int(not torch.equal(C._kernel_fft(plan, k), kf0))
        bad["conv"] += int(not torch.equal(C._conv(plan, u, kf0, None, None, False), y0))
        du, dk = dk_once()
        bad["du"] += int(not torch.equal(du, du0)); bad["dk"] += int(not torch.equal(dk, dk0))
    ref = ref_fft_conv(u, k, n=N)
    err = ((y0.double() - ref.double()).norm() / ref.double().norm()).item()
    res[f"N{N}"] = {"nondeterministic_launches_of_40": bad, "fwd_rel_err_vs_oracle": round(err, 5)}
print("RESULT " + json.dumps({"lib": os.environ.get("FFC_LIB", "product"), "res": res}))
