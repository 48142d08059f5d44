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
        os.environ["FFC_PERSIST"] = persist
        t = ev(lambda: C._conv(plan, u, kf, None, None, False))
        print(f"N={N} B={B} H={H} L={L} FFC_PERSIST={persist}: conv fwd {t:.4f} ms", flush=True)
    os.environ.pop("FFC_PERSIST")
