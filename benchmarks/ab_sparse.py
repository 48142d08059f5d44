"""FrequencySparseFFTConv: the compute-skipping forward kernel (ffc_conv_fwd_sparse) against the dense kernel on the same masked
k_f: forward ms at B16 H768 and bitwise comparison.  usage: python benchmarks/ab_sparse.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
from flashfftconv.sparse_conv import FrequencySparseFFTConv
lib, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
def ev(fn, it=20, rep=3):
    best = 1e9
    for _ in range(rep):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / it)
    return best
for (L, frac) in ((16384, 4), (16384, 8), (8192, 4), (16384, 32)):
    N, B, H = 2 * L, 16, 768
    x = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda") * 0.05
    m = FrequencySparseFFTConv(N // frac).cuda().eval()
    with torch.no_grad():
        y1 = m(x, k)
        conv = m._conv_for(x, keep=m.N_partial // 2)
        plan = conv._get_plan(x.device, conv._plan_seqlen)
        rows = C._sparse_rows(conv, plan)
        kf = C._kernel_fft(plan, k); kf.mul_(conv._kf_mask(plan, kf.dtype)[None, :, None])
        y = torch.empty_like(x)
        dense = lambda: _lib.check(lib.ffc_conv_fwd(plan.handle, P(x), P(kf), None, None, P(y), B, H, L, 0, sp()), "fwd")
        dense(); yd = y.clone()
        t_d = ev(dense)
        if rows:
            sparse = lambda: _lib.check(lib.ffc_conv_fwd_sparse(plan.handle, P(x), P(kf), None, None, P(y), B, H, L, 0, rows, sp()), "fwd_sp")
            sparse(); ys = y.clone()
            t_s = ev(sparse)
            print(f"FrequencySparseFFTConv(N/{frac}) L={L} fft={N}: rows per side {rows}: dense kernel {t_d:.4f} ms -> sparse kernel {t_s:.4f} ms ({(1 - t_s / t_d) * 100:.1f} % faster), "
                  f"bitwise equal {torch.equal(yd, ys)}, module output == sparse kernel {torch.equal(y1, ys)}", flush=True)
        else:
            print(f"FrequencySparseFFTConv(N/{frac}) L={L} fft={N}: no sparse kernel for this mask (dense {t_d:.4f} ms)", flush=True)
