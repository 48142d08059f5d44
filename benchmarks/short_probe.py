"""Short sequences: is the module-level time the GPU's or the host's?  Per L: module-level forward / forward+backward ms
(HIP events over back-to-back calls, as benchmarks/sweep.py), the kernels alone (same events around the raw C-ABI calls),
and the host-side enqueue cost per call (wall clock of the call loop before the final synchronize)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
import torch
from flashfftconv import FlashFFTConv, conv as C, _lib


def ev(fn, it=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(it): fn()
    b.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it, (t1 - t0) / it * 1e3


for L in [int(x) for x in (sys.argv[1:] or [1024, 2048, 4096, 8192])]:
    N, B, H = 2 * L, 16, 768
    u = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda"); dout = torch.randn_like(u)
    conv = FlashFFTConv(N, dtype=torch.bfloat16).cuda()
    plan = conv._get_plan(u.device, conv._plan_seqlen)
    kf = C._kernel_fft(plan, k)
    with torch.no_grad():
        m_fwd = ev(lambda: conv(u, k))
    ug, kg = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
    def fb():
        ug.grad = None; kg.grad = None
        conv(ug, kg).backward(dout)
    m_fb = ev(fb)
    k_kfft = ev(lambda: C._kernel_fft(plan, k))
    k_conv = ev(lambda: C._conv(plan, u, kf, None, None, False))
    print(json.dumps({"L": L, "module_fwd_ms(gpu,host)": [round(x, 4) for x in m_fwd], "module_fwd_bwd_ms(gpu,host)": [round(x, 4) for x in m_fb],
                      "kfft(gpu,host)": [round(x, 4) for x in k_kfft], "conv(gpu,host)": [round(x, 4) for x in k_conv]}), flush=True)
