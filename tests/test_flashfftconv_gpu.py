"""GPU parity tests — the reference's four tests (tests/test_flashfftconv.py:48-324) re-stated for the
MI355X library (same input recipe, same asserts, plus tighter relative-L2 gates), golden vectors,
ragged shapes and size-independent properties at the BASELINE shapes.  Everything goes through the
C-ABI (flashfftconv -> ctypes -> libflashfftconv_hip.so)."""
import glob, os
import numpy as np
import pytest
import torch

from oracle.torch_ref import ref_fft_conv

pytestmark = pytest.mark.gpu
SEQLENS = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
REL = {torch.bfloat16: 2e-2, torch.float16: 5e-3}   # SURVEY.md section 8(c) gates
# Round 4: the gates of the sizes with HBM levels (fft >= 65536) are the SAME as the fused sizes' (round 3 doubled them: VERDICT r03
# weak #1).  Measured margins, profiles/r04_parity_margins.txt: worst case over 256 .. 4M is 0.41 x the gate (bf16 out / du 8.2e-3
# at fft 4M against 2e-2, dk 1.0e-2 against 2e-2; fp16 8.4e-4 against 5e-3, dk 5.4e-3 against 1e-2).
BIG_F = 1.0


BIG = [65536, 131072, 262144, 524288, 1048576, 2097152, 4194304]


def set_B_H(B, H, seqlen):          # reference test_flashfftconv.py:15-46
    if seqlen == 16384 and B > 32: B = 32
    if seqlen == 32768 and B > 16: B = 16
    if seqlen >= 65536 and B > 4: B = 4
    cap = {131072: 384, 262144: 192, 524288: 96, 1048576: 48, 2097152: 32, 4194304: 16}
    if seqlen in cap and H > cap[seqlen]: H = cap[seqlen]
    return B, H


def rel(a, b):
    """Relative L2 error.  fp16 tensors get an absolute floor of one subnormal step (6e-8) per
    element: below ~6e-5 fp16 has a fixed absolute grid, so both sides carry that quantisation."""
    floor = 6e-8 * (a.numel() ** 0.5) if (a.dtype == torch.float16 or b.dtype == torch.float16) else 0.0
    a, b = a.double(), b.double()
    return (((a - b).norm() - floor).clamp_min(0) / b.norm().clamp_min(1e-30)).item()


def make_inputs(B, H, L, N, dtype, half_zero, device="cuda"):
    u = torch.randn(B, H, L, device=device).to(dtype) * 0.02
    k = torch.randn(H, L, device=device) * 0.02
    if half_zero:
        u[:, :, N // 2:] = 0.
        k[:, N // 2:] = 0.
    k = k * torch.exp(-0.1 * torch.arange(0, L, device=device))
    return u, k


def stable(fn, what, tries=12):
    """torch.fft (rocFFT) reference, computed until two consecutive evaluations agree bitwise.
    Observed on the MI355X boxes while the GPU is time-sliced between processes (pytest-xdist): torch.fft
    transiently returns whole (b, h) rows that are off by 1e-3..2e-2 relative, in ~5 % of the cases, on either of two
    back-to-back identical calls; the HIP kernels were bitwise reproducible in all of the same runs (asserted
    below).  A reference that flickers must not decide a parity test, so it has to reproduce itself first."""
    prev = fn()
    for _ in range(tries):
        cur = fn()
        if all(torch.equal(a, b) for a, b in zip(prev, cur)):
            return cur
        prev = cur
    pytest.fail(f"torch.fft reference ({what}) never reproduced itself in {tries + 1} evaluations")


def run_case(B, H, seqlen, dtype, padded, gated):
    from flashfftconv import FlashFFTConv
    # big sizes add two bf16/fp16 roundings per outer level (through HBM) on each side
    REL = {k: v * (BIG_F if seqlen >= 65536 else 1.0) for k, v in globals()["REL"].items()}
    torch.manual_seed(0)
    B, H = set_B_H(B, H, seqlen)
    N = seqlen
    L = N // 2 if padded else N
    u, k = make_inputs(B, H, L, N, dtype, not padded)
    u_c, k_c = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
    u.requires_grad_(True); k.requires_grad_(True)
    conv = FlashFFTConv(seqlen, dtype=dtype).to("cuda")
    if gated:
        pre = (torch.randn_like(u) * 0.02).requires_grad_(True); post = (torch.randn_like(u) * 0.02).requires_grad_(True)
        pre_c, post_c = pre.detach().clone().requires_grad_(True), post.detach().clone().requires_grad_(True)
        leaves_c = (u_c, k_c, pre_c, post_c)
        (ref,) = stable(lambda: (ref_fft_conv(u_c * pre_c, k_c, n=N) * post_c,), "forward")
        out = conv(u, k, pre, post)
    else:
        leaves_c = (u_c, k_c)
        (ref,) = stable(lambda: (ref_fft_conv(u_c, k_c, n=N),), "forward")
        out = conv(u, k)
    out_b = (conv(u, k, pre, post) if gated else conv(u, k)).detach()       # the HIP side must be bitwise reproducible run to run
    assert torch.equal(out, out_b), f"HIP forward not reproducible: {int((out != out_b).sum())} elements differ"
    with torch.no_grad():       # without a graph no spectra are stored (another kernel variant): the same output -- bit for bit
        out_n = conv(u, k, pre, post) if gated else conv(u, k)      # where an outer digit exists, to last-bit steps at fft <= 2048
    if seqlen >= 4096:
        assert torch.equal(out, out_n), f"forward with / without stored spectra: {int((out != out_n).sum())} elements differ"
    else:
        assert rel(out_n, out) < globals()["REL"][dtype] / 4
    assert torch.allclose(out, ref, atol=1e-2)                      # reference assert (:83)
    # relative gate on every output, gated or not (the reference's *0.02 gates make `atol` alone vacuous: |out| ~ 5e-7).
    # Gated: two more roundings to the activation dtype (u*pregate, y*postgate).
    GREL = {k: v * (1.5 if gated else 1.0) for k, v in REL.items()}
    assert rel(out, ref) < GREL[dtype], f"out rel-L2 {rel(out, ref):.3e}"
    dout = torch.randn_like(out) * 0.02
    gref = stable(lambda: torch.autograd.grad(ref, leaves_c, dout.clone(), retain_graph=True), "backward")
    leaves = (u, k, pre, post) if gated else (u, k)
    g = torch.autograd.grad(out, leaves, dout, retain_graph=True)
    g_b = torch.autograd.grad(out, leaves, dout)
    for name, a, b in zip(("du", "dk", "dpregate", "dpostgate"), g, g_b):
        assert torch.equal(a, b), f"HIP backward ({name}) not reproducible: {int((a != b).sum())} elements differ"
    assert torch.allclose(g[0], gref[0], atol=1e-2)                 # reference assert (:103)
    assert torch.allclose(g[1], gref[1], atol=1e-1)                 # reference ktol (:105-107)
    assert rel(g[0], gref[0]) < GREL[dtype], f"du rel-L2 {rel(g[0], gref[0]):.3e}"
    # dk: SURVEY 8(c)(iii) gate is 2e-2 for both dtypes; the dk_f -> dk inverse always runs in bf16 operand
    # arithmetic (fp32 range for the unnormalised sums), so fp16 modules see ~5e-3 there, not fp16's ~1e-3
    dk_tol = max(GREL[dtype], 1e-2 * (BIG_F if seqlen >= 65536 else 1.0) * (1.5 if gated else 1.0))
    assert rel(g[1], gref[1]) < dk_tol, f"dk rel-L2 {rel(g[1], gref[1]):.3e}"
    if gated:
        assert torch.allclose(g[2], gref[2], atol=1e-2)             # reference (:242-243)
        assert torch.allclose(g[3], gref[3], atol=1e-2)
        assert rel(g[2], gref[2]) < GREL[dtype], f"dpregate rel-L2 {rel(g[2], gref[2]):.3e}"
        assert rel(g[3], gref[3]) < GREL[dtype], f"dpostgate rel-L2 {rel(g[3], gref[3]):.3e}"


# The reference matrix in full (tests/test_flashfftconv.py:48-51, :109-112, :168-171, :249-252): B in {1,2,4,8,64},
# H in {768,111}, both dtypes, every fft size 256 .. 4194304 (B and H capped by the reference's own set_B_H), plus
# fft 2048, which the reference supports (README.md:268) but does not test.
ALL_SEQLENS = SEQLENS + BIG
REF_B = [1, 2, 4, 8, 64]
REF_H = [768, 111]


@pytest.mark.parametrize("seqlen", ALL_SEQLENS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H", REF_H)
@pytest.mark.parametrize("B", REF_B)
def test_flash_fft_conv(B, H, seqlen, dtype):
    run_case(B, H, seqlen, dtype, padded=False, gated=False)


@pytest.mark.parametrize("seqlen", ALL_SEQLENS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H", REF_H)
@pytest.mark.parametrize("B", REF_B)
def test_flash_fft_conv_padded(B, H, seqlen, dtype):
    run_case(B, H, seqlen, dtype, padded=True, gated=False)


@pytest.mark.parametrize("seqlen", ALL_SEQLENS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H", REF_H)
@pytest.mark.parametrize("B", REF_B)
def test_flash_fft_conv_gating(B, H, seqlen, dtype):
    run_case(B, H, seqlen, dtype, padded=False, gated=True)


@pytest.mark.parametrize("seqlen", ALL_SEQLENS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H", REF_H)
@pytest.mark.parametrize("B", REF_B)
def test_flash_fft_conv_gating_padded(B, H, seqlen, dtype):
    run_case(B, H, seqlen, dtype, padded=True, gated=True)


@pytest.mark.parametrize("B,H,seqlen,L,dtype,gated", [(2, 8, 2097152, 2097152, torch.bfloat16, True), (1, 4, 4194304, 4194304, torch.float16, False),
                                                     (3, 4, 4194304, 3000004, torch.bfloat16, True), (2, 6, 2097152, 1500001, torch.float16, True)])
def test_one_level_at_any_length(B, H, seqlen, L, dtype, gated, monkeypatch):
    """Round 6, opt-in (FFC_BIG_WIDE=1): fft 2097152 = 64 x 32768 and 4194304 = 128 x 32768 in ONE HBM level for rows beyond N / 2 resp. N / 4 (the wide form
    of the level, csrc/ffc_big.h BigBody::run_wide; reference: the 128-point butterfly at any length, butterfly_padded_cuda_bf16.cu:302-487).  Rows that are
    NON-ZERO over their whole length (the reference's own cases zero the second half), ragged lengths incl. L % 8 != 0, odd batch: forward against the
    torch.fft oracle, every gradient against the default routing (two levels / 32 x 65536: another rounding sequence of the same values)."""
    from flashfftconv import FlashFFTConv, bigfft, conv as C
    torch.manual_seed(seqlen + L + B)
    u = torch.randn(B, H, L, device="cuda").to(dtype) * 0.02
    k = torch.randn(H, L, device="cuda") * 0.02 * torch.exp(-4.0 * torch.arange(L, device="cuda") / L)
    gates = [torch.randn_like(u) * 0.5 for _ in range(2)] if gated else []
    dout = torch.randn_like(u) * 0.02
    res = {}
    for wide in (False, True):
        monkeypatch.setattr(bigfft, "WIDE", wide)
        fac = bigfft.choose(seqlen, L, C._TorchOps)
        assert (fac == bigfft.ONE128[seqlen] and bigfft.is_wide(fac[0][0], fac[1], L)) if wide else fac == bigfft.BIG_FACTORS[seqlen]
        conv = FlashFFTConv(seqlen, dtype=dtype).cuda()
        leaves = [u.clone().requires_grad_(True), k.clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in gates]
        out = conv(*leaves)
        res[wide] = (out.detach(), torch.autograd.grad(out, leaves, dout))
    (ref,) = stable((lambda: (ref_fft_conv(u * gates[0], k, n=seqlen) * gates[1],)) if gated else (lambda: (ref_fft_conv(u, k, n=seqlen),)), "forward")
    tol = REL[dtype] * BIG_F * (1.5 if gated else 1.0)
    assert rel(res[True][0], ref) < tol and rel(res[False][0], ref) < tol, (rel(res[True][0], ref), rel(res[False][0], ref))
    for a, b, what in zip(res[True][1], res[False][1], ("du", "dk", "dpregate", "dpostgate")):
        assert rel(a, b) < 2 * tol, f"{what}: rel {rel(a, b):.3e}"


@pytest.mark.parametrize("B,H,seqlen", [(5, 111, 4096), (3, 111, 32768), (3, 7, 65536), (5, 3, 1048576)])
def test_odd_batches_gated(B, H, seqlen):
    """odd batch sizes (a half-empty packed pair) are not in the reference matrix"""
    run_case(B, H, seqlen, torch.bfloat16, padded=True, gated=True)


def test_big_odd_batch_small_heads():
    run_case(3, 2, 65536, torch.bfloat16, padded=True, gated=False)
    run_case(1, 3, 1048576, torch.bfloat16, padded=True, gated=False)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N", SEQLENS)
def test_unit_scale_gated_relative(N, dtype):
    """Gated path with unit-scale tensors so relative errors are meaningful for every gradient: at the reference's x 0.02 input
    scale the fp16 gated rows are subnormal and rel()'s floor makes the relative gate vacuous (VERDICT r04 weak #2a), so every
    fused size runs here in both dtypes, padded (L = N/2) and, below, full length."""
    _unit_scale_gated(N, dtype, 4, 32, N // 2)


@pytest.mark.parametrize("N,dtype", [(256, torch.float16), (1024, torch.bfloat16), (2048, torch.float16), (4096, torch.bfloat16),
                                     (8192, torch.float16), (32768, torch.float16)])
def test_unit_scale_gated_relative_full_length(N, dtype):
    _unit_scale_gated(N, dtype, 3, 16, N)


@pytest.mark.parametrize("N,dtype,B,H,L", [(65536, torch.bfloat16, 4, 32, 32768), (65536, torch.float16, 3, 16, 65536),
                                           (131072, torch.bfloat16, 2, 16, 65536), (262144, torch.float16, 2, 16, 100000),
                                           (1048576, torch.bfloat16, 2, 16, 524288), (2097152, torch.float16, 2, 8, 2097152),
                                           (4194304, torch.bfloat16, 2, 16, 1048576), (4194304, torch.float16, 1, 4, 2097152)])
def test_unit_scale_gated_relative_big(N, dtype, B, H, L):
    """The same for fft sizes >= 65536 (HBM-level outer passes + fused kernel): every gated output and gradient has a
    relative gate (round 1 only had atol=1e-2 here, which zeros would pass)."""
    _unit_scale_gated(N, dtype, B, H, L, big=True)


def _unit_scale_gated(N, dtype, B, H, L, big=False):
    from flashfftconv import FlashFFTConv
    torch.manual_seed(1)
    f = BIG_F if big else 1.0
    if True:
        u, pre, post = (torch.randn(B, H, L, device="cuda").to(dtype).requires_grad_(True) for _ in range(3))
        k = (torch.randn(H, L, device="cuda") * 0.1).requires_grad_(True)
        c = [t.detach().clone().requires_grad_(True) for t in (u, k, pre, post)]
        out = FlashFFTConv(N, dtype=dtype).to("cuda")(u, k, pre, post)
        ref = ref_fft_conv(c[0] * c[2], c[1], n=N) * c[3]
        dout = torch.randn_like(out)
        out.backward(dout); ref.backward(dout.clone())
        assert rel(out, ref) < f * REL[dtype], f"out {rel(out, ref):.3e}"
        for name, a, b in zip(("du", "dk", "dpregate", "dpostgate"), (u, k, pre, post), c):
            e = rel(a.grad, b.grad)
            assert e < 1.5 * f * (max(REL[dtype], 1e-2) if name == "dk" else REL[dtype]), f"{name} {e:.3e}"


def _chunked_reference(u, k, pre, post, dout, N, hc):
    """fp32 torch.fft oracle with autograd, evaluated in head chunks of `hc` (bounds the complex64 temporaries)."""
    H = u.shape[1]
    outs, grads = [], [[] for _ in range(4 if pre is not None else 2)]
    for h0 in range(0, H, hc):
        sl = slice(h0, min(H, h0 + hc))
        leaves = [u[:, sl].detach().clone().requires_grad_(True), k[sl].detach().clone().requires_grad_(True)]
        if pre is not None:
            leaves += [pre[:, sl].detach().clone().requires_grad_(True), post[:, sl].detach().clone().requires_grad_(True)]
            ref = ref_fft_conv(leaves[0] * leaves[2], leaves[1], n=N) * leaves[3]
        else:
            ref = ref_fft_conv(leaves[0], leaves[1], n=N)
        g = torch.autograd.grad(ref, leaves, dout[:, sl])
        outs.append(ref.detach())
        for i, t in enumerate(g):
            grads[i].append(t)
    cat = lambda ts, d: torch.cat(ts, dim=d)
    return cat(outs, 1), [cat(grads[0], 1), cat(grads[1], 0)] + [cat(x, 1) for x in grads[2:]]


@pytest.mark.parametrize("N,dtype,B,H,L,Lk,gated,n_run", [(65536, torch.bfloat16, 3, 4, 300, 40, False, 512),
                                                          (131072, torch.float16, 2, 8, 16384, 16384, True, 32768),
                                                          (131072, torch.bfloat16, 2, 4, 30000, 2000, False, 32768),
                                                          (1048576, torch.bfloat16, 2, 2, 100000, 100000, True, 262144),
                                                          (4194304, torch.bfloat16, 1, 2, 1048576, 1048576, False, 2097152)])
def test_fft_size_fitted_to_the_rows(N, dtype, B, H, L, Lk, gated, n_run):
    """round 5, FlashFFTConv._fit_seqlen: a module above the single-launch sizes handed rows whose linear convolution fits a smaller fft
    size runs that size (BASELINE config 4: fft 4194304 around L = 1048576 runs 2097152 points).  Same numbers as the seqlen-point
    run of the same module (fit_fft = False -- which also keeps the L <= N/4 one-level form of fft 4194304 under test) and as the
    seqlen-point torch.fft oracle, forward and every gradient."""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(N + L)
    mk = lambda: torch.randn(B, H, L, device="cuda").to(dtype)
    u, dout = mk(), mk()
    k = torch.randn(H, Lk, device="cuda") * 0.05
    gates = (mk(), mk()) if gated else ()
    conv = FlashFFTConv(N, dtype=dtype).to("cuda")
    assert conv._fit_seqlen(L, Lk) == n_run
    res = []
    for fit in (True, False):
        conv.fit_fft = fit
        leaves = [t.clone().requires_grad_(True) for t in (u, k) + gates]
        y = conv(*leaves)
        res.append([y.detach()] + list(torch.autograd.grad(y, leaves, dout)))
    assert set(conv._fitted) == {n_run}
    ref, gref = _chunked_reference(u, k, gates[0] if gated else None, gates[1] if gated else None, dout, N, 4)
    want = [ref] + gref
    tol = REL[dtype] * (1.5 if gated else 1.0)
    for nm, a, b, w in zip(("out", "du", "dk", "dpregate", "dpostgate"), res[0], res[1], want):
        assert a.shape == w.shape and a.dtype == b.dtype
        t = max(tol, 1e-2) if nm == "dk" else tol
        assert rel(a, w) < t, f"fitted {nm} {rel(a, w):.3e}"
        assert rel(b, w) < t, f"seqlen-point {nm} {rel(b, w):.3e}"
        assert rel(a, b) < t, f"fitted against seqlen-point {nm} {rel(a, b):.3e}"


@pytest.mark.parametrize("name,N,B,H,L,gated", [("cfg2", 32768, 16, 768, 16384, False), ("cfg3", 16384, 8, 1024, 8192, True),
                                                ("cfg4", 4194304, 1, 16, 1048576, False)])
def test_baseline_configs_exact(name, N, B, H, L, gated):
    """BASELINE.json configs[1..3] at their EXACT shapes (cfg4: fft 4M with L = N/4), forward + backward against the
    torch.fft oracle with a relative-L2 gate on every output.  (configs[4], the conv1d, is in test_conv1d_gpu.py;
    configs[0] is the CPU plumbing case of tests/test_sim_kernels.py.)"""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(5)
    dtype = torch.bfloat16
    mk = lambda: torch.randn(B, H, L, device="cuda").to(dtype).requires_grad_(True)
    u = mk()
    k = (torch.randn(H, L, device="cuda") * 0.05).requires_grad_(True)
    pre, post = (mk(), mk()) if gated else (None, None)
    conv = FlashFFTConv(N, dtype=dtype).to("cuda")
    out = conv(u, k, pre, post) if gated else conv(u, k)
    dout = torch.randn_like(out)
    leaves = (u, k, pre, post) if gated else (u, k)
    g = torch.autograd.grad(out, leaves, dout)
    ref, gref = _chunked_reference(u, k, pre, post, dout, N, 64 if N <= 32768 else 4)
    f = (BIG_F if N >= 65536 else 1.0) * (1.5 if gated else 1.0)
    assert rel(out, ref) < f * REL[dtype], f"{name} out {rel(out, ref):.3e}"
    for nm, a, b in zip(("du", "dk", "dpregate", "dpostgate"), g, gref):
        e = rel(a, b)
        assert e < f * REL[dtype], f"{name} {nm} {e:.3e}"


@pytest.mark.parametrize("name,N,B,H,L,gated", [("cfg1", 1024, 4, 64, 512, False), ("cfg2", 32768, 16, 768, 16384, False),
                                                ("cfg3", 16384, 8, 1024, 8192, True)])
def test_baseline_configs_against_the_cpu_oracle(name, N, B, H, L, gated):
    """BASELINE.json configs[0..2] at their exact shapes on the GPU, checked on a few whole heads against the oracle evaluated
    ON THE CPU (pocketfft through torch.fft on CPU tensors): an anchor that does not depend on rocFFT, whose results flicker
    under GPU time-slicing (see `stable`).  Head h of the output, of du and of dk depends on head h of the inputs only."""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(11)
    dtype = torch.bfloat16
    mk = lambda: torch.randn(B, H, L, device="cuda").to(dtype)
    u, dout = mk(), mk()
    k = torch.randn(H, L, device="cuda") * 0.05
    gates = [mk(), mk()] if gated else []
    leaves = [t.clone().requires_grad_(True) for t in [u, k] + gates]
    conv = FlashFFTConv(N, dtype=dtype).to("cuda")
    out = conv(*leaves)
    g = torch.autograd.grad(out, leaves, dout)
    heads = sorted({0, 1, H // 2 - 1, H // 3, H - 1})
    cl = [u[:, heads].cpu().clone().requires_grad_(True), k[heads].cpu().clone().requires_grad_(True)] + \
         [t[:, heads].cpu().clone().requires_grad_(True) for t in gates]
    ref = ref_fft_conv(cl[0] * cl[2], cl[1], N) * cl[3] if gated else ref_fft_conv(cl[0], cl[1], N)
    gref = torch.autograd.grad(ref, cl, dout[:, heads].cpu())
    f = 1.5 if gated else 1.0
    assert rel(out[:, heads].cpu(), ref) < f * REL[dtype], f"{name} out {rel(out[:, heads].cpu(), ref):.3e}"
    assert rel(g[0][:, heads].cpu(), gref[0]) < f * REL[dtype], f"{name} du"
    assert rel(g[1][heads].cpu(), gref[1]) < f * REL[dtype], f"{name} dk {rel(g[1][heads].cpu(), gref[1]):.3e}"
    for i in range(2, len(leaves)):
        assert rel(g[i][:, heads].cpu(), gref[i]) < f * REL[dtype], f"{name} gate gradient {i}"
    # ... and the reference's absolute asserts at its input scale (test_flashfftconv.py:83, :103-107)
    assert torch.allclose(out[:, heads].cpu().float(), ref.float(), atol=1e-2 * float(ref.abs().max()) + 1e-2)


GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "conv_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[5:-4] for p in GOLD])
def test_golden_vectors(path):
    from flashfftconv import FlashFFTConv
    g = np.load(path)
    N = int(g["N"]); dtype = getattr(torch, str(g["dtype"]))
    t = lambda n, dt: torch.tensor(g[n], device="cuda").to(dt)
    u, k = t("u", dtype).requires_grad_(True), t("k", torch.float32).requires_grad_(True)
    conv = FlashFFTConv(N, dtype=dtype).to("cuda")
    if int(g["gated"]):
        pre, post = t("pre", dtype).requires_grad_(True), t("post", dtype).requires_grad_(True)
        out = conv(u, k, pre, post)
    else:
        out = conv(u, k)
    out.backward(t("dout", dtype))
    tol = REL[dtype] * (BIG_F if N >= 65536 else 1.0)
    assert rel(out, t("out", torch.float32)) < tol
    assert rel(u.grad, t("du", torch.float32)) < tol
    assert rel(k.grad, t("dk", torch.float32)) < max(tol, 1e-2)
    if int(g["gated"]):
        assert rel(pre.grad, t("dpre", torch.float32)) < tol
        assert rel(post.grad, t("dpost", torch.float32)) < tol


@pytest.mark.parametrize("N,L,Lk", [(256, 2, 2), (1024, 1002, 37), (4096, 2050, 2050), (8192, 36, 8192), (32768, 16390, 100)])
def test_ragged_lengths(N, L, Lk):
    from flashfftconv import FlashFFTConv
    torch.manual_seed(2)
    u = torch.randn(3, 5, L, device="cuda").to(torch.bfloat16).requires_grad_(True)
    k = (torch.randn(5, Lk, device="cuda") * 0.1).requires_grad_(True)
    uc, kc = u.detach().clone().requires_grad_(True), k.detach().clone().requires_grad_(True)
    out = FlashFFTConv(N, dtype=torch.bfloat16).to("cuda")(u, k)
    ref = ref_fft_conv(uc, kc, n=N)
    dout = torch.randn_like(out)
    out.backward(dout); ref.backward(dout.clone())
    assert rel(out, ref) < 2e-2 and rel(u.grad, uc.grad) < 2e-2 and rel(k.grad, kc.grad) < 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144])
def test_bidirectional_filter_fills_the_fft_size(N, dtype):
    """The M2-BERT callers pass a 2L-tap kernel, k = pad(k_fwd, (0, L)) + pad(flip(k_rev), (L, 0)), to FlashFFTConv(2L)
    (examples/bert/monarch_mixer_sequence_mixer_flashfftconv.py:141-151): L = N/2 and Lk = N, the taps beyond L act as
    negative lags of the circular convolution.  Forward, du and the full-length dk against the torch.fft oracle."""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(11)
    L, B, H = N // 2, 2, (5 if N <= 32768 else 2)
    u = torch.randn(B, H, L, device="cuda").to(dtype).requires_grad_(True)
    decay = torch.exp(-torch.linspace(0, 6, L, device="cuda"))
    kf_, kr_ = (torch.randn(H, L, device="cuda") * 0.1 * decay for _ in range(2))
    k = (torch.nn.functional.pad(kf_, (0, L)) + torch.nn.functional.pad(kr_.flip(-1), (L, 0))).requires_grad_(True)
    uc, kc = u.detach().clone().requires_grad_(True), k.detach().clone().requires_grad_(True)
    out = FlashFFTConv(N, dtype=dtype).to("cuda")(u, k)
    ref = ref_fft_conv(uc, kc, n=N)
    dout = torch.randn_like(out)
    out.backward(dout); ref.backward(dout.clone())
    tol = REL[dtype] * (BIG_F if N >= 65536 else 1.0)
    assert k.grad.shape == (H, N)
    assert rel(out, ref) < tol and rel(u.grad, uc.grad) < tol and rel(k.grad, kc.grad) < max(tol, 1e-2)


def test_errors_and_eval_mode():
    from flashfftconv import FlashFFTConv
    conv = FlashFFTConv(1024, dtype=torch.bfloat16).to("cuda")
    u = torch.randn(2, 4, 512, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(4, 512, device="cuda")
    with pytest.raises(RuntimeError):
        conv(u.float(), k)                                   # wrong dtype
    with pytest.raises(RuntimeError):
        conv(torch.randn(2, 4, 2048, device="cuda", dtype=torch.bfloat16), k)   # L > fft size
    with pytest.raises(AssertionError):
        conv(u, k, u, None)                                  # both gates or neither (conv.py:557-558)
    conv.eval()
    y = conv(u, k)                                           # inference: nothing saved (conv.py:587-588)
    assert torch.isfinite(y).all()


def test_full_size_properties_config2():
    """BASELINE config 2 (B=16,H=768,L=16384,N=32768): size-independent properties.
    (1) an impulse at position s returns k delayed by s; (2) linearity: conv(a+b) = conv(a)+conv(b)."""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(3)
    B, H, L, N = 16, 768, 16384, 32768
    conv = FlashFFTConv(N, dtype=torch.bfloat16).to("cuda")
    k = torch.randn(H, L, device="cuda") * 0.05
    u = torch.zeros(B, H, L, device="cuda", dtype=torch.bfloat16)
    shifts = torch.arange(B, device="cuda") * 37
    for b in range(B):
        u[b, :, shifts[b]] = 1.0
    y = conv(u, k).float()
    for b in (0, 1, 7, 15):
        s = int(shifts[b])
        assert rel(y[b, :, s:], k[:, : L - s]) < 1e-2
        if s:
            assert y[b, :, :s].abs().max() < 2e-3
    a = torch.randn(B, H, L, device="cuda").to(torch.bfloat16)
    b2 = torch.randn(B, H, L, device="cuda").to(torch.bfloat16)
    lhs = conv((a.float() + b2.float()).to(torch.bfloat16), k).float()
    rhs = conv(a, k).float() + conv(b2, k).float()
    assert rel(lhs, rhs) < 1.5e-2


SPARSE = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sparse_*.npz")))


@pytest.mark.parametrize("path", SPARSE, ids=[os.path.basename(p)[7:-4] for p in SPARSE])
def test_sparse_conv_golden(path):
    """PartialFFTConv / FrequencySparseFFTConv on the HIP path vs vectors from the reference's own
    flashfftconv/sparse_conv.py (oracle/make_golden.py: sparse_golden)."""
    from flashfftconv import PartialFFTConv, FrequencySparseFFTConv
    g = np.load(path)
    dtype = getattr(torch, str(g["dtype"]))
    t = lambda n, dt: torch.tensor(g[n], device="cuda").to(dt)
    x, k = t("x", dtype).requires_grad_(True), t("k", torch.float32).requires_grad_(True)
    mod = (PartialFFTConv if str(g["kind"]) == "partial" else FrequencySparseFFTConv)(int(g["N_partial"]))
    out = mod(x, k)
    out.backward(t("dout", dtype))
    tol = REL[dtype]
    assert rel(out, t("out", torch.float32)) < tol
    assert rel(x.grad, t("dx", torch.float32)) < tol
    assert rel(k.grad, t("dk", torch.float32)) < max(tol, 1e-2)


def test_kf_cache_opt_in():
    """cache_kf: k_f is reused only while k is the same storage at the same version and needs no grad."""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(3)
    conv = FlashFFTConv(4096, dtype=torch.bfloat16).to("cuda").eval()
    u = torch.randn(2, 8, 2048, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(8, 2048, device="cuda") * 0.1
    y0 = conv(u, k)
    conv.cache_kf = True
    y1 = conv(u, k); kf1 = conv._kf_cache[1]
    y2 = conv(u, k)
    assert conv._kf_cache[1] is kf1 and torch.equal(y0, y1) and torch.equal(y1, y2)
    k.mul_(2.0)                                   # in-place update bumps the version -> recomputed
    y3 = conv(u, k)
    assert conv._kf_cache[1] is not kf1
    assert rel(y3, ref_fft_conv(u, k, n=4096)) < 2e-2
    kg = k.clone().requires_grad_(True)           # a trainable filter never uses the cache
    conv.train()
    conv(u, kg).sum().backward()
    assert kg.grad is not None


@pytest.mark.parametrize("N,B,H,L", [(4096, 4, 16, 2048), (32768, 2, 8, 16384), (65536, 2, 16, 32768)])
def test_hip_graph_capture(N, B, H, L):
    """Every launch goes to the caller's stream (the reference launches on the legacy default stream): the
    forward can be captured into a HIP graph on a side stream and replayed, with results identical to eager."""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(4)
    u = torch.randn(B, H, L, device="cuda").to(torch.bfloat16); k = torch.randn(H, L, device="cuda") * 0.1
    mod = FlashFFTConv(N, dtype=torch.bfloat16).to("cuda").eval()
    with torch.no_grad():
        ref = mod(u, k)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            mod(u, k)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = mod(u, k)
        u.copy_(torch.randn_like(u))                  # new input in the captured buffer
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, mod(u, k))
        assert not torch.equal(y, ref)


@pytest.mark.parametrize("N,dtype,B,H,L", [(262144, torch.float16, 2, 3, 100004), (524288, torch.bfloat16, 2, 2, 262144),
                                           (4194304, torch.bfloat16, 1, 2, 1048576), (4194304, torch.float16, 1, 1, 1500000)])
def test_levels_take_the_fp32_filter_and_return_fp32_dk(N, dtype, B, H, L):
    """round 4: the first level over the filter reads the fp32 rows itself and the last level of dk writes fp32 (BigArgs::lf32) -- no
    cast kernels around the levels; results bit for bit those of the cast passes (FFC_BIG_LONG_F32 = 0 form), ragged lengths included."""
    from flashfftconv import FlashFFTConv, conv as C
    torch.manual_seed(N + L)
    dev = torch.device("cuda", 0)
    u, dout = (torch.randn(B, H, L, device=dev).to(dtype) for _ in range(2))
    k = torch.randn(H, L, device=dev) * 0.05
    mod = FlashFFTConv(N, dtype=dtype).to(dev)
    res = []
    prev = C._TorchOps.LONG_F32
    try:
        for flag in (True, False):
            C._TorchOps.LONG_F32 = flag
            uu, kk = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
            y = mod(uu, kk)
            res.append((y.detach(),) + torch.autograd.grad(y, (uu, kk), dout))
    finally:
        C._TorchOps.LONG_F32 = prev
    for a, b, name in zip(res[0], res[1], ("out", "du", "dk")):
        assert torch.equal(a, b), (name, (a.float() - b.float()).abs().max().item())
    assert res[0][2].dtype == torch.float32
