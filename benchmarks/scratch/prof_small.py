"""fft 4096 forward/backward timing vs the FFC_PERSIST grid cap (tuning aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (N, B, H, L) in ((4096, 16, 768, 2048), (4096, 16, 768, 4096), (4096, 16, 12288, 4096)):
    u = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    for persist in ("0", "256", "512", "128"):
        os.environ["FFC_PERSIST"] = persist; __import__("flashfftconv.conv").conv.reload_env()
        t = ev(lambda: C._conv(plan, u, kf, None, None, False))
        print(f"N={N} B={B} H={H} L={L} FFC_PERSIST={persist}: conv fwd {t:.4f} ms", flush=True)
    os.environ.pop("FFC_PERSIST"); __import__("flashfftconv.conv").conv.reload_env()
from flashfftconv import _lib
lib = _lib.lib(); sp = _lib.stream_ptr
for (N, B, H, L) in ((4096, 16, 768, 2048), (8192, 16, 768, 4096), (16384, 16, 768, 8192)):
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    du = torch.empty_like(u); dk = torch.empty(H, L, dtype=torch.float32, device="cuda")
    t = {"kfft": ev(lambda: C._kernel_fft(plan, k)),
         "fwd": ev(lambda: C._conv(plan, u, kf, None, None, False)),
         "bwd_fused": ev(lambda: _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(du), None, _lib.ptr(ws), B, H, L, sp()), "bwd")),
         "dkifft": ev(lambda: _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, L, _lib.ptr(dk), sp()), "dk"))}
    print(f"N={N} L={L} ws={ws.numel()/1e6:.0f} MB: " + "  ".join(f"{a} {b:.4f}" for a, b in t.items()), flush=True)
