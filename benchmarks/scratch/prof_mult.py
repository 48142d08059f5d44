"""FFC_WG_MULT (workgroups per CU targeted by ffc_choose_chunks) sweep, same process."""
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
for (N, B, H, L) in ((32768, 16, 768, 16384), (16384, 8, 1024, 8192), (16384, 16, 768, 8192), (8192, 16, 768, 4096), (4096, 16, 768, 2048), (32768, 64, 768, 16384)):
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k); du = torch.empty_like(u); dk = torch.empty(H, L, device="cuda")
    for mult in ("1", "2", "3", "4", "8"):
        os.environ["FFC_WG_MULT"] = mult; __import__("flashfftconv.conv").conv.reload_env()
        ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
        tf = ev(lambda: C._conv(plan, u, kf, None, None, False))
        def bw():
            _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(du), None, _lib.ptr(ws), B, H, L, sp()), "bwd")
            _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, L, _lib.ptr(dk), sp()), "dk")
        tb = ev(bw)
        print(f"N={N} B={B} H={H} L={L} FFC_WG_MULT={mult}: conv_fwd {tf:.4f}  bwd+dkifft {tb:.4f}  ws {ws.numel()/1e6:.0f} MB", flush=True)
    os.environ.pop("FFC_WG_MULT"); __import__("flashfftconv.conv").conv.reload_env()
