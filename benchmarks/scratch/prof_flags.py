"""A/B of the FFC_FLAGS tuning bits on the fused backward / forward of config 2 (2: k_f streamed, 4: scratch streamed)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib(); sp = _lib.stream_ptr
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (N, B, H, L) in ((32768, 16, 768, 16384), (16384, 16, 768, 8192)):
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda"); du = torch.empty_like(u)
    for rep in range(2):
        for fl in ("0", "2", "4", "6"):
            os.environ["FFC_FLAGS"] = fl; __import__("flashfftconv.conv").conv.reload_env()
            tf = ev(lambda: C._conv(plan, u, kf, None, None, False))
            tb = ev(lambda: _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(du), None, _lib.ptr(ws), B, H, L, sp()), "bwd"))
            print(f"N={N} FFC_FLAGS={fl}: conv_fwd {tf:.4f}  bwd_fused {tb:.4f}", flush=True)
    os.environ.pop("FFC_FLAGS"); __import__("flashfftconv.conv").conv.reload_env()
