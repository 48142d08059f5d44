"""BASELINE sweep: FlashFFTConv fwd, bwd, fwd+bwd for B=16, H=768, L in {1K..1M} (fft size 2L), plus the
five BASELINE.json configs.  Where memory forces fewer heads the time is rescaled linearly to H=768
(the reference does the same, benchmarks/benchmark_flashfftconv.py:28-59,111) and the row says so.
Prints one JSON line per row.  Method mirrors reference benchmarks/benchmark.py (separate fwd / bwd
timings, warm-up, mean over repeats) but with HIP events."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
import torch
from flashfftconv import FlashFFTConv, FlashDepthWiseConv1d


REPEATS = 3


def ev_time(fn, iters, repeats=REPEATS):
    """HIP events around `iters` back-to-back calls, `repeats` times after 5 warm-up calls: (median, min) ms per call."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def peak_bytes(fn):
    """peak device memory of one call on top of what is allocated before it (inputs, parameters, plan tables):
    torch.cuda.max_memory_allocated as reference benchmarks/benchmark.py:137-147 measures it, minus the resident base"""
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    fn()
    torch.cuda.synchronize()
    return int(torch.cuda.max_memory_allocated() - base)


def _torch_fft_conv(u, k, N, g):
    """the torch.fft form the reference compares its memory against (tests/test_flashfftconv.py:5-13; README.md:232)"""
    x = u * g[0] if g else u
    y = torch.fft.ifft(torch.fft.fft(x.float(), n=N) * torch.fft.fft(k.float(), n=N), n=N).real.to(u.dtype)[..., : u.shape[-1]]
    return y * g[1] if g else y


def peak_mem_row(mod, u, k, g, dout, N):
    """peak memory (bytes above the resident inputs) of: the inference forward; forward + backward with the module default
    (save_spectrum on: the forward keeps FFT(u) for the backward pass); the same with save_spectrum off (the reference's
    footprint: inputs only); the torch.fft form.  README.md:232 publishes the ratio torch.fft / FlashFFTConv ("memory savings")."""
    leaves = [u, k] + list(g)

    def clear():
        for t in leaves:
            t.grad = None

    def fb(fn):
        def run():
            clear()
            fn().backward(dout)
        return run
    out = {}
    with torch.no_grad():
        mod.eval(); out["fwd_infer"] = peak_bytes(lambda: mod(u, k, *g)); mod.train()
    keep = mod.save_spectrum
    out["fwd_bwd_save_spectrum"] = peak_bytes(fb(lambda: mod(u, k, *g)))
    mod.save_spectrum = False
    out["fwd_bwd_recompute"] = peak_bytes(fb(lambda: mod(u, k, *g)))
    mod.save_spectrum = keep
    try:
        with torch.no_grad():
            out["fwd_torch_fft"] = peak_bytes(lambda: _torch_fft_conv(u, k, N, g))
        out["fwd_bwd_torch_fft"] = peak_bytes(fb(lambda: _torch_fft_conv(u, k, N, g)))
        out["saving_vs_torch_fft_fwd"] = round(out["fwd_torch_fft"] / max(out["fwd_infer"], 1), 2)
        out["saving_vs_torch_fft_fwd_bwd"] = round(out["fwd_bwd_torch_fft"] / max(out["fwd_bwd_save_spectrum"], 1), 2)
    except torch.cuda.OutOfMemoryError:
        out["fwd_bwd_torch_fft"] = None
    clear()
    out["inputs_bytes"] = int(sum(t.numel() * t.element_size() for t in leaves) + dout.numel() * dout.element_size())
    out["how"] = "max_memory_allocated over one call minus memory_allocated before it (bytes; H_run heads, not rescaled)"
    return out


def set_iters(N):
    return 20 if N < 1048576 else 8 if N == 1048576 else 5


def conv_row(name, N, B, H, L, dtype=torch.bfloat16, gated=False, Hrun=None):
    Hrun = Hrun or H
    dev = "cuda"
    u = torch.randn(B, Hrun, L, device=dev).to(dtype).requires_grad_(True)
    k = torch.randn(Hrun, L, device=dev).requires_grad_(True)
    g = [torch.randn(B, Hrun, L, device=dev).to(dtype).requires_grad_(True) for _ in range(2)] if gated else []
    dout = torch.randn(B, Hrun, L, device=dev).to(dtype)
    mod = FlashFFTConv(N, dtype=dtype).to(dev)
    iters = set_iters(N)
    from flashfftconv import conv as _C
    fb0 = dict(_C.SPECTRUM_FALLBACKS)      # a timed row that silently fell back to the recomputing backward says so
    # forward = the TRAINING forward (grad enabled: it also stores what the backward pass reads, e.g. the spectra of
    # module.save_spectrum); the inference forward (no_grad, eval) is reported next to it
    (t_f, t_f_min) = ev_time(lambda: mod(u, k, *g), iters)
    with torch.no_grad():
        mod.eval()
        (t_fi, _) = ev_time(lambda: mod(u, k, *g), iters)
        mod.train()
    y = mod(u, k, *g)
    leaves = [u, k] + g

    def bwd():
        for t in leaves:
            t.grad = None          # otherwise autograd adds an accumulate pass over every gradient tensor
        y.backward(dout, retain_graph=True)
    (t_b, t_b_min) = ev_time(bwd, iters)
    del y
    # short sequences: the eager backward above is bound by the host side of autograd, not by its kernels; the
    # whole fwd+bwd step as ONE HIP graph (FlashFFTConv.graphed_step) is what the GPU actually needs for it
    t_g = None
    if N <= 8192:
        step = mod.graphed_step(u, k, dout, *g)
        (t_g, _) = ev_time(step.replay, iters)
        del step
    # a module that ran a smaller fft size than it was built for (FlashFFTConv._fit_seqlen, config 4): the seqlen-point run next to it
    unfit = None
    if mod._fit_seqlen(L, L) != N:
        mod.fit_fft = False
        (tu_f, _) = ev_time(lambda: mod(u, k, *g), iters)
        y = mod(u, k, *g)
        (tu_b, _) = ev_time(bwd, iters)
        del y
        mod.fit_fft = True
        unfit = (tu_f, tu_b)
    fb = {k_: _C.SPECTRUM_FALLBACKS[k_] - fb0[k_] for k_ in fb0 if _C.SPECTRUM_FALLBACKS[k_] != fb0[k_]}
    pm = peak_mem_row(mod, u, k, g, dout, N)
    scale = H / Hrun
    t_f, t_b, t_fi, t_f_min, t_b_min = t_f * scale, t_b * scale, t_fi * scale, t_f_min * scale, t_b_min * scale
    rows = B * H
    lg = math.log2(N)
    fft_f, fft_b = 2 * 5 * N * lg + 6 * N, 3 * 5 * N * lg + 14 * N
    alg_f = B * H * L * 2 * (2 + 2 * gated) + H * N * 4
    alg_b = B * H * L * 2 * (3 + 4 * gated) + H * N * 4 + H * N * 8
    return ({"row": name, "fft": N, "B": B, "H": H, "L": L, "dtype": str(dtype).split(".")[-1], "gated": gated,
                      "H_run": Hrun, "rescaled": Hrun != H,
                      "fwd_ms": round(t_f, 4), "bwd_ms": round(t_b, 4), "fwd_bwd_ms": round(t_f + t_b, 4),
                      "fwd_ms_min": round(t_f_min, 4), "bwd_ms_min": round(t_b_min, 4), "fwd_infer_ms": round(t_fi, 4),
                      "timing": f"median (and min) of {REPEATS} x {iters} iterations, HIP events",
                      "seq_per_s": round(rows / ((t_f + t_b) * 1e-3)),
                      "tflops_fft_equiv": round(rows * (fft_f + fft_b) / ((t_f + t_b) * 1e-3) / 1e12, 2),
                      "fwd_alg_GBs": round(alg_f / (t_f * 1e-3) / 1e9), "bwd_alg_GBs": round(alg_b / (t_b * 1e-3) / 1e9),
                      "fwd_hbm_frac": round(alg_f / (t_f * 1e-3) / 8e12, 4), "bwd_hbm_frac": round(alg_b / (t_b * 1e-3) / 8e12, 4),
                      # fft size the module ran: the smallest one that holds the rows' linear convolution (FlashFFTConv._fit_seqlen)
                      "fft_run": mod._fit_seqlen(L, L),
                      **({"spectrum_fallbacks": fb} if fb else {}),
                      **({"fwd_ms_seqlen_points": round(unfit[0] * scale, 4), "bwd_ms_seqlen_points": round(unfit[1] * scale, 4)} if unfit else {}),
                      "peak_mem_bytes": pm, **({"graph_step_ms": round(t_g * scale, 4)} if t_g is not None else {})})


def conv1d_row():
    import torch.nn as nn
    B, D, L, K = 64, 2048, 8192, 3
    ref = nn.Conv1d(D, D, K, groups=D, padding=1).cuda()
    m = FlashDepthWiseConv1d(D, K, 1, ref.weight.detach(), ref.bias.detach(), is_bhl=True, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(B, D, L, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    with torch.no_grad():
        (t_f, _) = ev_time(lambda: m(x), 20)
    y = m(x)
    dout = torch.randn_like(y)
    def bwd():
        x.grad = None
        for prm in m.parameters():
            prm.grad = None
        y.backward(dout, retain_graph=True)
    (t_b, _) = ev_time(bwd, 20)
    byts = B * L * D * 2 * 2 + K * D * 2
    return ({"row": "cfg5 conv1d k=3 B=64 H=2048 L=8192 bf16 BHL", "fwd_ms": round(t_f, 4), "bwd_ms": round(t_b, 4),
                      "fwd_GBs": round(byts / (t_f * 1e-3) / 1e9), "bwd_GBs": round(1.5 * byts / (t_b * 1e-3) / 1e9),
                      "fwd_hbm_frac": round(byts / (t_f * 1e-3) / 8e12, 4), "bwd_hbm_frac": round(1.5 * byts / (t_b * 1e-3) / 8e12, 4)})


def config_rows():
    yield conv_row("cfg2 FlashFFTConv(32768) B16 H768 L16384", 32768, 16, 768, 16384)
    yield conv_row("cfg3 gated fft16384 B8 H1024 L8192", 16384, 8, 1024, 8192, gated=True)
    yield conv_row("cfg4 FlashFFTConv(4194304) B1 H16 L1048576", 4194304, 1, 16, 1048576)
    yield conv1d_row()


def sweep_rows(lg_lo=10, lg_hi=20):
    """BASELINE metric: B=16, H=768, L = 1K .. 1M, fft size 2L (heads reduced and rescaled above fft 128K, as the
    reference's own benchmark does)"""
    for lg in range(lg_lo, lg_hi + 1):
        L = 1 << lg
        N = 2 * L
        Hrun = 768 if N <= 131072 else max(16, 768 * 131072 // N)
        yield conv_row(f"sweep L={L}", N, 16, 768, L, Hrun=Hrun)
        torch.cuda.empty_cache()


def sweep_gated_rows(lgs=(10, 12, 14, 16)):
    """gated, padded (fft = 2L) fwd + bwd at B=16 H=768: the reference's benchmark grid has these forms next to the plain ones
    (benchmarks/benchmark_flashfftconv.py:150-212); a subset of the lengths keeps the default bench run short"""
    for lg in lgs:
        L = 1 << lg
        yield conv_row(f"gated sweep L={L}", 2 * L, 16, 768, L, gated=True)
        torch.cuda.empty_cache()


# The reference's published table (README.md:224-230): gated forward, fp16, L = N, time scaled to
# B = 64 x H = 768 rows, 1 x H100-SXM.  Same shapes here (B, H shrunk as the reference's set_B_H does and rescaled the same
# way, benchmarks/benchmark_flashfftconv.py:28-59, :111), same training-mode forward.
H100_GATED_FWD_MS = {256: 0.11, 1024: 0.29, 4096: 1.43, 8192: 3.58, 16384: 12.2, 32768: 26.3, 1048576: 1768.9, 2097152: 4623.5,
                     4194304: 10049.4}


def _ref_B_H(seqlen, B=64, H=768):
    if seqlen == 16384: B = min(B, 32)
    if seqlen == 32768: B = min(B, 16)
    if seqlen >= 65536: B = min(B, 8)
    cap = {131072: 384, 262144: 192, 524288: 96, 1048576: 48, 2097152: 32, 4194304: 16}
    return B, min(H, cap.get(seqlen, H))


def readme_rows(sizes=None):
    for N in (sizes or sorted(H100_GATED_FWD_MS)):
        B, H = _ref_B_H(N)
        u = torch.randn(B, H, N, device="cuda").to(torch.float16).requires_grad_(True)
        k = torch.randn(H, N, device="cuda").requires_grad_(True)
        g = [torch.randn(B, H, N, device="cuda").to(torch.float16).requires_grad_(True) for _ in range(2)]
        mod = FlashFFTConv(N, dtype=torch.float16).cuda()
        (t, tmin) = ev_time(lambda: mod(u, k, *g), set_iters(N))
        # the reference's own tool and call (benchmarks/benchmark.py:8-24: torch.utils.benchmark.Timer(...).timeit(repeats), mean): host time
        # around `repeats` calls with one synchronisation at the end, so it includes whatever the host adds when the launches are short
        import torch.utils.benchmark as tbench
        t_timer = tbench.Timer(stmt="mod(u, k, *g)", globals={"mod": mod, "u": u, "k": k, "g": g},
                               num_threads=torch.get_num_threads()).timeit(set_iters(N)).mean * 1e3
        with torch.no_grad():       # the same forward without what the training forward stores for the backward pass
            mod.eval()
            (ti, _) = ev_time(lambda: mod(u, k, *g), set_iters(N))
            mod.train()
        adj = 64 * 768 / (B * H)
        dout = torch.randn(B, H, N, device="cuda").to(torch.float16)
        # the gated backward at the same shape (reference benchmarks/benchmark_flashfftconv.py:150-212 times fwd, bwd and memory
        # of the gated forms; its README publishes the forward only)
        y = mod(u, k, *g)
        leaves = [u, k] + g

        def bwd():
            for t_ in leaves:
                t_.grad = None
            y.backward(dout, retain_graph=True)
        (tb, _) = ev_time(bwd, set_iters(N))
        del y
        for t_ in leaves:
            t_.grad = None
        pm = peak_mem_row(mod, u, k, g, dout, N)
        del dout
        yield {"row": f"README table N={N}", "peak_mem_bytes": pm, "fft": N, "L": N, "dtype": "float16", "gated": True, "B_run": B, "H_run": H,
               "fwd_ms_scaled_to_B64_H768": round(t * adj, 3), "fwd_ms_min_scaled": round(tmin * adj, 3),
               "fwd_ms_timer_scaled": round(t_timer * adj, 3), "fwd_no_grad_ms_scaled": round(ti * adj, 3), "bwd_ms_scaled": round(tb * adj, 3),
               "h100_ms_published": H100_GATED_FWD_MS[N], "speedup_vs_h100_published": round(H100_GATED_FWD_MS[N] / (t * adj), 2),
               "speedup_vs_h100_timer": round(H100_GATED_FWD_MS[N] / (t_timer * adj), 2)}
        del u, k, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "row":       # one shape: sweep.py row N B H L [H_run] [gated]  (routing A/B runs: FFC_MULTIPASS=... in the environment)
        N, B, H, L = (int(x) for x in sys.argv[2:6])
        Hrun = int(sys.argv[6]) if len(sys.argv) > 6 else None
        r = conv_row(f"fft {N} B{B} H{H} L{L} FFC_MULTIPASS={os.environ.get('FFC_MULTIPASS', 'default')}", N, B, H, L, Hrun=Hrun, gated=len(sys.argv) > 7)
        print(json.dumps({k: v for k, v in r.items() if k != "peak_mem_bytes"} | {"peak_fwd_bwd": r["peak_mem_bytes"]["fwd_bwd_save_spectrum"]}), flush=True)
    if which in ("all", "configs"):
        for r in config_rows():
            print(json.dumps(r), flush=True)
    if which in ("all", "sweep"):
        for r in sweep_rows():
            print(json.dumps(r), flush=True)
    if which in ("all", "gated"):
        for r in sweep_gated_rows():
            print(json.dumps(r), flush=True)
    if which in ("all", "readme"):
        for r in readme_rows():
            print(json.dumps(r), flush=True)
