"""Kernel logic on CPU: the SAME kernel body the GPU runs (ffc_body.h / ffc_modes.h) is executed by
the 64-lane wave simulator (csrc/ffc_sim.cpp) and compared with the oracle."""
import ctypes
import numpy as np
import pytest

import simlib as S
from oracle import ref_fft_conv as O

SIZES = [256, 512, 1024, 4096, 8192, 16384, 32768]
TOL = {0: 1.2e-2, 1: 1.5e-3}     # rel-L2 gates: bf16, fp16
NAME = {0: "bf16", 1: "fp16"}


def rel(a, b):
    a = np.asarray(a)
    a = a.astype(np.complex128) if np.iscomplexobj(a) else a.astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def q(x, dt):
    return S.from_bits(S.to_bits(x, dt), dt).astype(np.float64)


@pytest.mark.parametrize("N", SIZES)
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("padded", [False, True])
def test_conv_fwd(N, dt, padded):
    rng = np.random.default_rng(N + dt)
    L = N // 2 if padded else N
    B, H = 3, 2                      # odd batch: last pair has an empty imaginary lane
    u = rng.standard_normal((B, H, L)).astype(np.float32)
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)          # k -> k_f through the simulated kfft kernel
    y = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf), dt)
    assert rel(y, O.ref_fft_conv(q(u, dt), k, N)) < TOL[dt]


@pytest.mark.parametrize("N,B", [(8192, 5), (8192, 10), (16384, 5), (16384, 2)])
def test_conv_cross_unit_groups(N, B):
    """fft 8192 / 16384: phase B runs tile tau of two units (two pairs of a head) in lock-step (Body::inner_tile2x) and falls
    back to two tiles of one unit when a group's partner unit has no pair left: pair counts 1, 3, 5 leave every combination
    (both active / first only / group idle) somewhere in the job loop.  Gated, padded, both k_f signs."""
    rng = np.random.default_rng(N + B)
    dt, H, L = 0, 2, N // 2
    u, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(3))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.make_kf_internal(k, N, dt)
    y = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, S.to_bits(g1, dt), S.to_bits(g2, dt)), dt)
    assert rel(y, O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype=NAME[dt])) < TOL[dt]
    du = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, conj=1), dt)
    dref, _ = O.ref_grads(np.zeros_like(u), k, q(u, dt), N)
    assert rel(du, dref) < TOL[dt]


@pytest.mark.parametrize("N", [256, 1024, 4096, 32768])
@pytest.mark.parametrize("dt", [0, 1])
def test_conv_gated_and_conj(N, dt):
    rng = np.random.default_rng(7 * N + dt)
    L, B, H = N // 2, 2, 2
    u, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(3))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.make_kf_internal(k, N, dt)
    y = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, S.to_bits(g1, dt), S.to_bits(g2, dt)), dt)
    ref = O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype=NAME[dt])
    assert rel(y, ref) < TOL[dt]
    # conj(k_f): the input-gradient pass
    du = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, conj=1), dt)
    dref, _ = O.ref_grads(np.zeros_like(u), k, q(u, dt), N)
    assert rel(du, dref) < TOL[dt]


@pytest.mark.parametrize("N,L", [(256, 2), (256, 250), (1024, 1002), (4096, 2050), (8192, 36), (32768, 16390),
                                 (16384, 5001), (16384, 8200), (16384, 16383)])      # fft 16384: the forward with the folded outer twiddle
def test_conv_ragged_lengths(N, L):
    """L not a multiple of 8 takes the element-wise I/O path; tiny and odd-ish lengths included."""
    rng = np.random.default_rng(L)
    u = rng.standard_normal((2, 2, L)).astype(np.float32)
    k = (rng.standard_normal((2, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, 0, k)
    y = S.from_bits(S.sim_conv_fwd(N, 0, S.to_bits(u, 0), kf), 0)
    assert rel(y, O.ref_fft_conv(q(u, 0), k, N)) < TOL[0]


def test_linearity_and_shift_properties():
    """Size-independent properties: conv is linear in u and commutes with a delay."""
    N, L, dt = 4096, 2048, 0
    rng = np.random.default_rng(3)
    k = (rng.standard_normal((1, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    imp = np.zeros((2, 1, L), np.float32); imp[0, 0, 0] = 1.0; imp[1, 0, 5] = 1.0
    y = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(imp, dt), kf), dt)
    assert rel(y[0, 0], k[0].astype(np.float64)) < TOL[dt]                 # identity impulse -> k
    assert rel(y[1, 0, 5:], k[0, :-5].astype(np.float64)) < TOL[dt]        # delayed impulse -> delayed k
    assert np.abs(y[1, 0, :5]).max() < 1e-3


@pytest.mark.parametrize("N", SIZES)
@pytest.mark.parametrize("dt", [0, 1])
def test_kernel_fft(N, dt):
    rng = np.random.default_rng(N)
    H, Lk = 5, N // 2 + 4
    k = (rng.standard_normal((H, Lk)) * 0.1).astype(np.float32)
    got = S.from_bits(S.sim_kernel_fft(N, dt, k), dt)
    got = got[..., 0] + 1j * got[..., 1]
    nt, sf, sk, freq = S.plan_info(N, dt)
    ref = np.fft.fft(k.astype(np.float64), n=N, axis=-1)[:, freq] * sk
    assert rel(got, ref) < TOL[dt]


@pytest.mark.parametrize("N,L,B,H,nch,gated", [(256, 128, 20, 2, 2, True), (512, 512, 3, 2, 1, False), (1024, 512, 6, 3, 1, True),
                                               (4096, 4096, 18, 1, 2, True), (8192, 4096, 3, 2, 1, False),
                                               (16384, 16384, 5, 1, 2, False), (32768, 16384, 4, 2, 2, True)])
@pytest.mark.parametrize("dt", [0, 1])
def test_dk_path(N, L, B, H, nch, gated, dt):
    """dk_f accumulation (fp32 slabs, several chunks) + inverse -> dk, vs analytic gradient."""
    rng = np.random.default_rng(N + B)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    Lk = L - 4
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    dk = S.sim_dk(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), Lk, pre, post, nchunk=nch)
    v = q(q(u, dt) * q(g1, dt), dt) if gated else q(u, dt)
    dc = q(q(d, dt) * q(g2, dt), dt) if gated else q(d, dt)
    ref = np.fft.ifft((np.fft.fft(dc, n=N) * np.conj(np.fft.fft(v, n=N))).sum(0)).real[:, :Lk]
    assert rel(dk, ref) < 1.5 * TOL[0]       # the dk inverse always runs in bf16 arithmetic (fp32 range)


def test_golden_forward_through_simulator():
    """Committed golden vectors (reference oracle outputs) vs the simulated kernels."""
    import glob, os
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "conv_N*_plain.npz")))[:6]:
        g = np.load(path)
        N = int(g["N"]); dt = 0 if str(g["dtype"]) == "bfloat16" else 1
        k = g["k"]
        if N == 2048:     # flashfftconv/conv.py FOLDED_SEQLENS: fft 2048 runs on the 4096 plan with k periodised
            kp = np.zeros((k.shape[0], 2048), np.float32); kp[:, :k.shape[1]] = k
            k, N = np.concatenate([kp, kp], -1), 4096
        kf = S.sim_kernel_fft(N, dt, k)
        y = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(g["u"], dt), kf), dt)
        assert rel(y, g["out"].astype(np.float64)) < TOL[dt] * 1.5, path
        assert np.allclose(y, g["out"], atol=1e-2)      # the reference's own assert (test_flashfftconv.py:83)


@pytest.mark.parametrize("N,L,B,gated", [(65536, 32768, 2, False), (131072, 131072, 1, True), (262144, 100004, 3, False),
                                         (2097152, 600000, 1, True)])      # 2M: one outer level (32) around the 2-pass fft 65536
def test_big_sizes_through_outer_levels(N, L, B, gated):
    """FFT sizes >= 65536: HBM-level outer passes (csrc/ffc_big.h) + fused inner kernel, orchestrated by
    flashfftconv/bigfft.py on the simulator backend; forward (gated / ragged / odd batch) and dk."""
    from flashfftconv import bigfft as BG
    rng = np.random.default_rng(N)
    dt, H = 0, 1
    ops = S.SimOps()
    u, g1, g2, d = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.05).astype(np.float32)
    ub, g1b, g2b, db = (S.to_bits(x, dt) for x in (u, g1, g2, d))
    M = BG.BIG_FACTORS[N][1]
    kf = BG.kernel_fft(ops, dt, N, k, H, L)
    x = BG.levels_forward(ops, dt, N, ub, B, H, L, g1b if gated else None)
    y = ops.conv(dt, M, x, kf, False)
    out = np.zeros_like(ub)
    BG.levels_inverse(ops, dt, N, y, out, B, H, L, g2b if gated else None)
    ref = O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype="bf16") if gated else O.ref_fft_conv(q(u, dt), k, N)
    assert rel(S.from_bits(out, dt), ref) < 1.5e-2
    xu = BG.levels_forward(ops, dt, N, ub, B, H, L)
    xd = BG.levels_forward(ops, dt, N, db, B, H, L)
    dk = BG.dk_from_slabs(ops, N, ops.dkf(dt, M, xd, xu), xu.shape[0], H, L)
    _, dkref = O.ref_grads(q(u, dt), k, q(d, dt), N)
    assert rel(dk, dkref) < 1.5e-2


@pytest.mark.parametrize("N,L,rows", [(32768, 16384, 4), (32768, 32768, 2), (16384, 8192, 4), (16384, 5000, 1)])
def test_frequency_sparse_kernel_skips_zero_rows(N, L, rows):
    """ffc_conv_fwd_sparse (kernel variant SP): with a low-pass k_f (non-zero bins |f| < rows N / 32) the compute-skipping
    kernel returns bit for bit what the dense kernel returns on the same masked k_f, forward and conj(k_f) pass (fft 16384: to
    rounding, the dense kernel there folds its outer twiddle)."""
    rng = np.random.default_rng(N + rows)
    dt, B, H = 0, 3, 1
    u, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(3))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kfn = np.fft.fft(k.astype(np.float64), n=N)
    f = np.arange(N)
    kfn[:, (f >= rows * N // 32) & (f <= N - rows * N // 32)] = 0          # keep |f| < rows * N / 32
    kf = S.make_kf_from_spectrum(kfn, N, dt)
    ub = S.to_bits(u, dt)
    for conj in (0, 1):
        dense = S.sim_conv_fwd(N, dt, ub, kf, S.to_bits(g1, dt), S.to_bits(g2, dt), conj=conj)
        S.lib().ffcsim_set_sparse(rows)
        try:
            sparse = S.sim_conv_fwd(N, dt, ub, kf, S.to_bits(g1, dt), S.to_bits(g2, dt), conj=conj)
        finally:
            S.lib().ffcsim_set_sparse(0)
        if N == 16384:      # round 5: the dense forward of fft 16384 folds its outer twiddle into the stage matrices, the sparse variant keeps the chains
            assert rel(S.from_bits(sparse, dt), S.from_bits(dense, dt).astype(np.float64)) < 1e-2
        else:
            assert np.array_equal(dense, sparse), f"conj={conj}: {int((dense != sparse).sum())} elements differ"
    yref = np.fft.ifft(np.fft.fft(q(u, dt).astype(np.float64) * q(g1, dt), n=N) * kfn[None], n=N).real[..., :L] * q(g2, dt)
    S.lib().ffcsim_set_sparse(rows)
    try:
        y = S.from_bits(S.sim_conv_fwd(N, dt, ub, kf, S.to_bits(g1, dt), S.to_bits(g2, dt)), dt)
    finally:
        S.lib().ffcsim_set_sparse(0)
    assert rel(y, yref) < 1.5 * TOL[dt]


@pytest.mark.parametrize("L,B,gated,f", [(131072, 2, False, 128), (100004, 1, True, 128), (131072, 1, True, 64), (77776, 3, False, 64)])
def test_one_level_of_128(L, B, gated, f):
    """the factor-128 level (4 passes of the 32-point outer kernel, ffc_outer_pass_r: how fft 4194304 = 128 x 32768 runs when
    L <= N / 4) on a size the simulator finishes: fft 524288 = 128 x 4096, L <= N / 4, forward (gated, ragged, odd batch) and dk."""
    from flashfftconv import bigfft as BG
    N, fac = f * 4096, ((f,), 4096)          # L <= N / (f / 32)
    rng = np.random.default_rng(L)
    dt, H, M = 0, 1, 4096
    ops = S.SimOps()
    u, g1, g2, d = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.05).astype(np.float32)
    ub, g1b, g2b, db = (S.to_bits(x, dt) for x in (u, g1, g2, d))
    kf = BG.kernel_fft(ops, dt, N, k, H, L, fac)
    x = BG.levels_forward(ops, dt, N, ub, B, H, L, g1b if gated else None, fac)
    assert x.shape == (2 * ((B + 1) // 2), H * f, M)
    # one launch for all passes (ffc_outer_pass_all) == one launch per pass (ffc_outer_pass_r): the forward bit for bit
    per_pass = S.SimOps(); per_pass.one_launch = False
    assert np.array_equal(x, BG.levels_forward(per_pass, dt, N, ub, B, H, L, g1b if gated else None, fac))
    y = ops.conv(dt, M, x, kf, False)
    out = np.zeros_like(ub)
    BG.levels_inverse(ops, dt, N, y, out, B, H, L, g2b if gated else None, None, fac)
    ref = O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype="bf16") if gated else O.ref_fft_conv(q(u, dt), k, N)
    assert rel(S.from_bits(out, dt), ref) < 1.5e-2
    xd = BG.levels_forward(ops, dt, N, db, B, H, L, None, fac)
    xu = BG.levels_forward(ops, dt, N, ub, B, H, L, None, fac)
    dk = BG.dk_from_slabs(ops, N, ops.dkf(dt, M, xd, xu), xu.shape[0], H, L, None, fac)
    _, dkref = O.ref_grads(q(u, dt), k, q(d, dt), N)
    assert rel(dk, dkref) < 1.5e-2


@pytest.mark.parametrize("L,B,gated,f", [(524288, 2, False, 128), (300004, 1, True, 128), (400008, 3, True, 128), (262144, 1, True, 64), (200001, 2, False, 64)])
def test_one_level_of_128_at_any_length(L, B, gated, f):
    """Round 6, the WIDE form of the factor-128 / 64 level (BigBody::run_wide: up to R * 32 long-side rows, an R-point butterfly of the row blocks in front
    of the pass matrices; how fft 4194304 = 128 x 32768 and 2097152 = 64 x 32768 run at L > N / 4 resp. N / 2) on sizes the simulator finishes:
    fft 524288 = 128 x 4096 and 262144 = 64 x 4096 with rows up to L = N -- forward (gated, ragged incl. L % 8 != 0, odd batch), k -> k_f and dk."""
    from flashfftconv import bigfft as BG
    N, fac = f * 4096, ((f,), 4096)
    assert L > N // (f // 32)      # beyond the first 32 long-side rows
    rng = np.random.default_rng(L)
    dt, H, M = 0, 1, 4096
    ops = S.SimOps()
    assert BG.is_wide(f, M, L)
    u, g1, g2, d = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.05 * np.exp(-np.arange(L) / (L / 6.0))).astype(np.float32)
    ub, g1b, g2b, db = (S.to_bits(x, dt) for x in (u, g1, g2, d))
    kf = BG.kernel_fft(ops, dt, N, k, H, L, fac)
    x = BG.levels_forward(ops, dt, N, ub, B, H, L, g1b if gated else None, fac)
    assert x.shape == (2 * ((B + 1) // 2), H * f, M)
    # the level against its definition: X_{k0}[m] = s W_N^{m k0} sum_{n0} x[n0 M + m] W_f^{n0 k0}, rows stored as c * 32 + d for k0 = c + R d
    R = f // 32
    xin = q(u, dt).astype(np.float64) * (q(g1, dt) if gated else 1.0)
    z = np.zeros((N,), np.complex128)
    z[:L] = xin[0, 0] + 1j * (xin[1, 0] if B > 1 else 0.0)
    X = np.fft.fft(z.reshape(f, M), axis=0) * np.exp(-2j * np.pi * np.outer(np.arange(f), np.arange(M)) / N) * BG.level_scale(f)
    order = [c + R * dd for c in range(R) for dd in range(32)]
    got = S.from_bits(x[0, :f], dt).astype(np.float64) + 1j * S.from_bits(x[1, :f], dt)
    assert rel(got, X[order]) < 6e-3, rel(got, X[order])
    y = ops.conv(dt, M, x, kf, False)
    out = np.zeros_like(ub)
    BG.levels_inverse(ops, dt, N, y, out, B, H, L, g2b if gated else None, None, fac)
    ref = O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype="bf16") if gated else O.ref_fft_conv(q(u, dt), k, N)
    assert rel(S.from_bits(out, dt), ref) < 1.5e-2, rel(S.from_bits(out, dt), ref)
    xd = BG.levels_forward(ops, dt, N, db, B, H, L, None, fac)
    xu = BG.levels_forward(ops, dt, N, ub, B, H, L, None, fac)
    dk = BG.dk_from_slabs(ops, N, ops.dkf(dt, M, xd, xu), xu.shape[0], H, L, None, fac)
    _, dkref = O.ref_grads(q(u, dt), k, q(d, dt), N)
    assert rel(dk, dkref) < 1.5e-2, rel(dk, dkref)


@pytest.mark.parametrize("N,L,B,H,nch,gated", [(256, 128, 9, 2, 1, True), (1024, 1024, 3, 2, 1, False), (4096, 2048, 5, 2, 2, True),
                                               (16384, 8192, 3, 1, 1, False), (32768, 16384, 3, 2, 2, True)])
@pytest.mark.parametrize("dt", [0, 1])
def test_fused_backward(N, L, B, H, nch, gated, dt):
    """Modes::bwd: du (+ dpregate) and the dk_f slabs from one pass (3 transforms per pair)."""
    rng = np.random.default_rng(N + L)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    du, dpre, dk = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, L, pre, post, nch)
    if gated:
        r = O.ref_grads(q(u, dt), k, q(d, dt), N, q(g1, dt), q(g2, dt))
        assert rel(S.from_bits(dpre, dt), r[2]) < TOL[dt]
        if N >= 4096:       # dpostgate out of the same pass (one extra inverse transform of the first spectrum)
            assert rel(S.from_bits(S.sim_bwd.dpost, dt), r[3]) < TOL[dt]
    else:
        r = O.ref_grads(q(u, dt), k, q(d, dt), N)
    assert rel(S.from_bits(du, dt), r[0]) < TOL[dt]
    assert rel(dk, r[1]) < 1.5 * TOL[0]


# ---------------------------------------------------------------- multi-pass sizes (fft 65536 / 131072 = 2 / 4 passes of
# the fused 32768 kernel, csrc/ffc_body.h struct Pass): every mode, padded / full / ragged lengths (1, 2 or 4 input blocks)
@pytest.mark.parametrize("N,L,B", [(65536, 32768, 3), (65536, 65536, 2), (65536, 16384, 2), (65536, 40004, 1), (65536, 1002, 2),
                                   (131072, 65536, 2), (131072, 131072, 1), (131072, 32768, 2), (131072, 100000, 1)])
@pytest.mark.parametrize("dt", [0, 1])
def test_multipass_forward(N, L, B, dt):
    rng = np.random.default_rng(N + L + dt)
    H = 2 if L <= 32768 else 1
    u, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(3))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)                       # kfft kernel, multi-pass
    assert rel(S.from_bits(kf, dt), S.from_bits(S.make_kf_internal(k, N, dt), dt).astype(np.float64)) < (8e-3 if dt == 0 else 1e-3)
    y = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf), dt)
    assert rel(y, O.ref_fft_conv(q(u, dt), k, N)) < TOL[dt]
    yg = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, S.to_bits(g1, dt), S.to_bits(g2, dt)), dt)
    assert rel(yg, O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype=NAME[dt])) < TOL[dt]
    du = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, conj=1), dt)      # conj(k_f): the input-gradient pass
    dref, _ = O.ref_grads(np.zeros_like(u), k, q(u, dt), N)
    assert rel(du, dref) < TOL[dt]


@pytest.mark.parametrize("N,L,B,H,nch,gated", [(65536, 32768, 3, 2, 2, True), (65536, 65536, 2, 1, 1, False), (65536, 8192, 5, 1, 1, False),
                                               (131072, 65536, 3, 1, 1, True), (131072, 131072, 2, 1, 1, False)])
@pytest.mark.parametrize("dt", [0, 1])
def test_multipass_backward(N, L, B, H, nch, gated, dt):
    rng = np.random.default_rng(N + L)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    du, dpre, dk = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, L, pre, post, nch)
    if gated:
        r = O.ref_grads(q(u, dt), k, q(d, dt), N, q(g1, dt), q(g2, dt))
        assert rel(S.from_bits(dpre, dt), r[2]) < TOL[dt]
        assert rel(S.from_bits(S.sim_bwd.dpost, dt), r[3]) < TOL[dt]
    else:
        r = O.ref_grads(q(u, dt), k, q(d, dt), N)
    assert rel(S.from_bits(du, dt), r[0]) < TOL[dt]
    assert rel(dk, r[1]) < 1.5 * TOL[0]
    # the dk_f-only kernel (Modes::dkf) + inverse, shorter dk than L
    Lk = L - 4
    dk2 = S.sim_dk(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), Lk, pre, post, nchunk=nch)
    assert rel(dk2, r[1][:, :Lk]) < 1.5 * TOL[0]


# ---------------------------------------------------------------- fft 2048 = 2 passes of the inner-only 32 x 32 kernel
@pytest.mark.parametrize("L,B,H", [(1024, 3, 2), (2048, 2, 2), (512, 5, 1), (1500, 2, 1), (2, 1, 1)])
@pytest.mark.parametrize("dt", [0, 1])
def test_fft2048_inner_multipass(L, B, H, dt):
    N = 2048
    rng = np.random.default_rng(L + dt)
    u, g1, g2, d = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    assert rel(S.from_bits(kf, dt), S.from_bits(S.make_kf_internal(k, N, dt), dt).astype(np.float64)) < (8e-3 if dt == 0 else 1e-3)
    y = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf), dt)
    assert rel(y, O.ref_fft_conv(q(u, dt), k, N)) < TOL[dt]
    yg = S.from_bits(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, S.to_bits(g1, dt), S.to_bits(g2, dt)), dt)
    assert rel(yg, O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype=NAME[dt])) < TOL[dt]
    for gated in (False, True):
        pre = S.to_bits(g1, dt) if gated else None
        post = S.to_bits(g2, dt) if gated else None
        du, dpre, dk = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, L, pre, post, 2)
        r = O.ref_grads(q(u, dt), k, q(d, dt), N, q(g1, dt), q(g2, dt)) if gated else O.ref_grads(q(u, dt), k, q(d, dt), N)
        assert rel(S.from_bits(du, dt), r[0]) < TOL[dt]
        assert rel(dk, r[1]) < 1.5 * TOL[0]
        if gated:
            assert rel(S.from_bits(dpre, dt), r[2]) < TOL[dt]
        dk2 = S.sim_dk(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), max(L - 4, 1), pre, post, nchunk=1)
        assert rel(dk2, r[1][:, :max(L - 4, 1)]) < 1.5 * TOL[0]


# ---------------------------------------------------------------- spectrum-saving pair (ffc_conv_fwd_z -> ffc_conv_bwd_z / _zy) and
# the LDS-DMA input rows of its backward (Body::rows_dma: next pair's dout rows copied into the dead half of the exchange buffer)
@pytest.mark.parametrize("N,L,B,H,nch,gated", [(32768, 16384, 5, 1, 1, False),      # 3 pairs per unit: prologue + 2 run-ahead copies, odd batch
                                               (32768, 16376, 4, 1, 1, False),      # ragged fast path: tail of the last row zeroed
                                               (32768, 9000, 2, 1, 1, False),       # rows beyond L (copied clamped, zeroed), partial row
                                               (8192, 4096, 10, 1, 1, False),       # 4 units per workgroup, 2 waves per unit, ragged unit count
                                               (8192, 4096, 6, 2, 2, True),         # gated: register path + dpost from the dout row load
                                               (16384, 8192, 4, 1, 1, True),        # 16-point outer digit: no DMA path, side product only
                                               (4096, 2048, 6, 1, 1, False)])
@pytest.mark.parametrize("dt", [0])
def test_saved_spectrum_backward_and_dma_rows(N, L, B, H, nch, gated, dt):
    rng = np.random.default_rng(N + L + B)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    ub, db = S.to_bits(u, dt), S.to_bits(d, dt)
    S.lib().ffcsim_dma_count.restype = ctypes.c_long
    S.lib().ffcsim_dma_count()
    y, du, dpre, dpost, ws = S.sim_fwd_bwd_z(N, dt, ub, db, kf, pre, post, nch, flags=0)
    ndma = S.lib().ffcsim_dma_count()
    y8, du8, dpre8, dpost8, ws8 = S.sim_fwd_bwd_z(N, dt, ub, db, kf, pre, post, nch, flags=8)      # 8: register path for the rows
    assert S.lib().ffcsim_dma_count() == 0
    # one copy per (plane of a pair, E row < 16, wave): 32-point outer digit, plain rows.  A straight-line burst since late round 4: rows
    # beyond L and the missing row of an odd batch are copied too (clamped) and zeroed by rows_dma_finish -- per-row skips had put a
    # s_waitcnt in front of every copy of the pair loop's prologue
    want = (((B + 1) // 2) * 2 * H * 16 * (N // 4096)) if (N in (8192, 32768) and not gated) else 0
    assert ndma == want, (ndma, want)
    assert np.array_equal(du, du8) and np.array_equal(ws, ws8), "LDS-DMA input rows change the result"
    # against the recomputing kernel: du bit for bit (same arithmetic on the same spectrum of dout), dk to the spectrum's rounding
    du0, dpre0, dk0 = S.sim_bwd(N, dt, db, ub, kf, L, pre, post, nch)
    assert np.array_equal(du, du0)
    nt, _, _, _ = S.plan_info(N, dt)
    dk = np.full((H, L), np.nan, np.float32)
    assert S.lib().ffcsim_kernel_ifft_grad(N, dt, S.p(ws), ws.size // (H * nt * 2048), H, L, S.p(dk)) == 0
    # fft 16384: the spectrum-saving forward folds its outer twiddle into the tile matrices, the recomputing backward does not -- the two
    # bf16 spectra differ by a rounding of the matrix entries as well (same gate as tests/test_spectrum_gpu.py)
    assert rel(dk, dk0.astype(np.float64)) < (1.2e-2 if N == 16384 else 3e-3)
    r = O.ref_grads(q(u, dt), k, q(d, dt), N, q(g1, dt), q(g2, dt)) if gated else O.ref_grads(q(u, dt), k, q(d, dt), N)
    assert rel(S.from_bits(du, dt), r[0]) < TOL[dt] and rel(dk, r[1]) < 1.5 * TOL[0]
    if gated:
        assert np.array_equal(dpre, dpre0)
        assert rel(S.from_bits(dpost, dt), r[3]) < TOL[dt]


# ---------------------------------------------------------------- single-tile sizes (fft <= 2048): the training forward that keeps the pair's
# spectrum and / or the output before the postgate (conv_kernel<.., SZ>, conv_rp_kernel<.., SZ>) and the backward on them.  Round 6: the gated
# form keeps the output ALONE (keep = "y": the backward transforms u * pregate again from rows it loads anyway) -- every result bit for bit
# what the recomputing kernels and the both-kept form give, dpostgate = dout * y_raw against the oracle
@pytest.mark.parametrize("N,L,B,H,gated", [(1024, 1024, 4, 2, True), (1024, 500, 3, 2, True), (256, 256, 9, 1, True), (256, 100, 5, 2, True),
                                           (2048, 1024, 4, 1, True), (2048, 2048, 3, 1, True), (2048, 1500, 2, 1, True), (1024, 1024, 4, 1, False),
                                           (2048, 1024, 3, 1, False)])
@pytest.mark.parametrize("dt", [0, 1])
def test_single_tile_kept_output_and_spectrum(N, L, B, H, gated, dt):
    rng = np.random.default_rng(N + L + B + dt)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    ub, db = S.to_bits(u, dt), S.to_bits(d, dt)
    y0 = S.sim_conv_fwd(N, dt, ub, kf, pre, post)
    du0, dpre0, dk0 = S.sim_bwd(N, dt, db, ub, kf, L, pre, post, 1)
    nt, _, _, _ = S.plan_info(N, dt)
    r = O.ref_grads(q(u, dt), k, q(d, dt), N, q(g1, dt), q(g2, dt)) if gated else O.ref_grads(q(u, dt), k, q(d, dt), N)
    for keep in (("zy", "y") if gated else ("zy",)):
        y, du, dpre, dpost, ws = S.sim_fwd_bwd_z(N, dt, ub, db, kf, pre, post, 1, keep=keep)
        assert np.array_equal(y, y0), keep
        assert np.array_equal(du, du0), keep
        dk = np.full((H, L), np.nan, np.float32)
        assert S.lib().ffcsim_kernel_ifft_grad(N, dt, S.p(ws), ws.size // (H * nt * 2048), H, L, S.p(dk)) == 0
        if keep == "y":
            assert np.array_equal(dk, dk0), "the recomputed spectrum is the recomputing kernel's"
        assert rel(dk, dk0.astype(np.float64)) < 3e-3 and rel(dk, r[1]) < 1.5 * TOL[0]
        if gated:
            assert np.array_equal(dpre, dpre0), keep
            assert rel(S.from_bits(dpost, dt), r[3]) < TOL[dt], keep


# ---------------------------------------------------------------- HBM-level outer pass: persistent double-buffered form (BigBody::run_pipe:
# next block's rows by LDS-DMA into the idle exchange buffer) against the one-block-per-workgroup form -- same arithmetic, bit for bit
@pytest.mark.parametrize("n0,mi,B,H,L,gated", [(32, 1024, 3, 2, 32768, False),     # 3 "workgroups" x 4 blocks, odd batch (missing Im row)
                                               (32, 512, 2, 3, 9000, False),        # rows beyond L skipped, a partial 1 KB piece zero-filled
                                               (16, 2048, 2, 1, 20008, False),      # 16-point level: two pieces per row
                                               (32, 512, 2, 2, 16384, True)])       # gated: forward keeps run<>, the inverse pipelines (gate at the store)
@pytest.mark.parametrize("dt", [0, 1])
def test_outer_pass_pipelined_equals_per_block(n0, mi, B, H, L, gated, dt):
    rng = np.random.default_rng(n0 + mi + L)
    npair = (B + 1) // 2
    x = S.to_bits(rng.standard_normal((B, H, L)).astype(np.float32), dt)
    gate = S.to_bits(rng.standard_normal((B, H, L)).astype(np.float32), dt) if gated else None
    L_ = S.lib()
    L_.ffcsim_dma_count.restype = ctypes.c_long
    res = {}
    for pipe in (1, 0):
        L_.ffcsim_set_big_pipe(pipe)
        try:
            L_.ffcsim_dma_count()
            mid = np.full((2 * npair, H * n0, mi), 0x7fc0 if dt == 0 else 0x7e00, np.uint16)       # NaN-filled: every element must be written
            assert L_.ffcsim_big_outer(n0, dt, 1, S.p(x), S.p(mid), S.p(gate), B, npair, H, mi, L, ctypes.c_float(0.125)) == 0
            nf = L_.ffcsim_dma_count()
            out = np.zeros_like(x)
            assert L_.ffcsim_big_outer(n0, dt, 0, S.p(mid), S.p(out), S.p(gate), B, npair, H, mi, L, ctypes.c_float(0.25)) == 0
            ni = L_.ffcsim_dma_count()
        finally:
            L_.ffcsim_set_big_pipe(0)
        res[pipe] = (mid, out, nf, ni)
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
    assert res[0][2] == 0 and res[0][3] == 0
    assert (res[1][2] > 0) == (not gated) and res[1][3] == 2 * npair * H * n0 * (mi * 2 // 1024)      # inverse: every 1 KB row piece once
    # forward then inverse of a level = identity x (n0 * 0.125 * 0.25) on the valid part
    ref = S.from_bits(x, dt).astype(np.float64) * (S.from_bits(gate, dt).astype(np.float64) ** 2 if gated else 1.0) * (n0 * 0.125 * 0.25)
    assert rel(S.from_bits(res[1][1], dt), ref) < (2e-2 if dt == 0 else 3e-3)


# ---------------------------------------------------------------- frequency-sparse convolution at the HBM-level sizes: bigfft.row_freq says
# which natural frequency every (inner k_f row, inner position) holds; checked against the spectrum the kernels really produce
@pytest.mark.parametrize("N,fac", [(65536, ((16,), 4096)), (131072, ((32,), 4096)), (1048576, ((16, 16), 4096)), (524288, ((128,), 4096)),
                                   (262144, ((64,), 4096))])      # (the inner size needs an outer digit: complex-input k -> k_f)
def test_row_freq_locates_every_bin_of_the_inner_rows(N, fac):
    from flashfftconv import bigfft as BG
    rng = np.random.default_rng(N + len(fac[0]))
    dt, H, L = 0, 1, N // (fac[0][0] // 32) if fac[0][0] > 32 else N // 2
    k = (rng.standard_normal((H, L)) * np.exp(-0.002 * np.arange(L))).astype(np.float32)
    kf = S.from_bits(BG.kernel_fft(S.SimOps(), dt, N, k, H, L, fac), dt).astype(np.float64)      # (rows, kf_elems, 2), unscaled K_f
    offs, stride = BG.row_freq(N, fac)
    nt, _, _, freq = S.plan_info(fac[1], dt)
    assert kf.shape[0] == len(offs) and kf.shape[1] == len(freq)
    K = np.fft.fft(S.from_bits(S.to_bits(k, dt), dt).astype(np.float64)[0], n=N)
    f = (np.asarray(offs)[:, None] + stride * np.asarray(freq)[None, :]) % N
    got = kf[..., 0] + 1j * kf[..., 1]
    assert rel(got, K[f]) < 2e-2


# ---------------------------------------------------------------- dk out of the backward launch itself (Modes::dk_tail): the workgroup that
# owns every pair of a head inverts its accumulation registers right away -- no fp32 slab, no dkifft launch
@pytest.mark.parametrize("N,L,B,H,gated,Lk", [(32768, 16384, 4, 2, False, 16384), (32768, 16384, 3, 1, True, 16384), (32768, 32768, 2, 1, False, 32768),
                                              (32768, 9000, 2, 2, False, 700), (32768, 16384, 2, 1, False, 16381),
                                              # several units per workgroup: the units' sums are added up through LDS first
                                              (16384, 8192, 5, 2, True, 8192), (16384, 16384, 2, 1, False, 16384), (8192, 4096, 9, 1, False, 4096),
                                              (8192, 3000, 3, 2, True, 77)])
@pytest.mark.parametrize("dt", [0, 1])
def test_dk_from_the_backward_launch(N, L, B, H, gated, Lk, dt):
    rng = np.random.default_rng(L + B + Lk)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, Lk)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    du0, dpre0, dk0 = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, Lk, pre, post, 1)
    du1, dpre1, dk1 = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, Lk, pre, post, 1, fused_dk=True)
    assert np.array_equal(du0, du1) and (not gated or np.array_equal(dpre0, dpre1))
    assert not np.isnan(dk1).any() and rel(dk1, dk0.astype(np.float64)) < 1e-3
    kpad = np.zeros((H, max(L, Lk)), np.float32); kpad[:, :Lk] = k
    r = O.ref_grads(q(u, dt), kpad[:, :L] if Lk <= L else kpad, q(d, dt), N, q(g1, dt), q(g2, dt)) if gated else O.ref_grads(q(u, dt), kpad[:, :L] if Lk <= L else kpad, q(d, dt), N)
    assert rel(dk1, r[1][:, :Lk]) < 1.5 * TOL[0]


# ... and at the multi-pass sizes (round 5, Modes::dk_tail_rp; bf16 plans): every pass of the backward kernel ends with its share of dk,
# added to the fp32 dk rows by the wave that wrote the earlier passes' sums -- no slab per (head, pass), no dkifft launch
@pytest.mark.parametrize("N,L,B,H,gated,Lk", [(65536, 32768, 3, 1, False, 32768), (65536, 65536, 2, 1, True, 65536), (65536, 20000, 2, 2, False, 333),
                                              (131072, 65536, 1, 1, False, 65536)])
def test_dk_from_the_multi_pass_backward_launch(N, L, B, H, gated, Lk):
    dt = 0
    rng = np.random.default_rng(L + B + Lk)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, Lk)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    du0, dpre0, dk0 = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, Lk, pre, post, 1)
    du1, dpre1, dk1 = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, Lk, pre, post, 1, fused_dk=True)
    assert np.array_equal(du0, du1) and (not gated or np.array_equal(dpre0, dpre1))
    # the same fp32 sums through the same inverse: the slab path rounds nothing in between, so the two agree to the last bit
    assert not np.isnan(dk1).any() and np.array_equal(dk1, dk0)


# ---------------------------------------------------------------- k -> k_f inside the forward launch (Modes::kfft_head, ConvArgs::kfuse_k)
@pytest.mark.parametrize("N,L,B,H,gated,Lk", [(32768, 16384, 4, 2, False, 16384), (32768, 32768, 2, 1, True, 32768), (32768, 9000, 3, 2, False, 700),
                                              (32768, 16384, 2, 1, False, 16381),
                                              (16384, 8192, 5, 2, True, 8192), (16384, 16384, 3, 1, False, 101), (8192, 4096, 10, 1, False, 4096), (8192, 8192, 5, 2, True, 8191)])
@pytest.mark.parametrize("dt", [0, 1])
def test_kernel_fft_inside_the_forward_launch(N, L, B, H, gated, Lk, dt):
    rng = np.random.default_rng(L + B + Lk)
    u, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(3))
    k = (rng.standard_normal((H, Lk)) * 0.1).astype(np.float32)
    pre = S.to_bits(g1, dt) if gated else None
    post = S.to_bits(g2, dt) if gated else None
    kf0 = S.sim_kernel_fft(N, dt, k)
    y0 = S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf0, pre, post)
    kf1 = np.full_like(kf0, 0x7fc0 if dt == 0 else 0x7e00)                   # NaN pattern: the launch has to write every k_f tile it reads
    kc = np.ascontiguousarray(k)
    S.lib().ffcsim_set_fused_k(S.p(kc), Lk)
    try:
        y1 = S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf1, pre, post)
    finally:
        S.lib().ffcsim_set_fused_k(None, 0)
    assert np.array_equal(kf0, kf1) and np.array_equal(y0, y1)


# ---------------------------------------------------------------- the same two fusions in their complex forms (inner convolution of the
# HBM-level sizes): k_f rows from a pair-plane tensor inside the forward launch, dk rows into a pair-plane tensor out of the backward launch
@pytest.mark.parametrize("M,B,H", [(32768, 2, 2), (16384, 4, 1), (16384, 2, 2)])
@pytest.mark.parametrize("dt", [0, 1])
def test_complex_forms_of_the_fused_launches(M, B, H, dt):
    rng = np.random.default_rng(M + B + dt)
    x, dd = (rng.standard_normal((B, H, M)).astype(np.float32) for _ in range(2))
    xk = S.to_bits(rng.standard_normal((2, H, M)).astype(np.float32) * 0.1, dt)
    nt, _, _, _ = S.plan_info(M, dt)
    L_ = S.lib()
    kf0 = np.zeros((H, nt * 1024, 2), np.uint16)
    assert L_.ffcsim_kernel_fft_c(M, dt, S.p(xk), H, S.p(kf0), ctypes.c_float(0.5)) == 0
    xb, db = S.to_bits(x, dt), S.to_bits(dd, dt)
    y0 = S.sim_conv_fwd(M, dt, xb, kf0)
    kf1 = np.full_like(kf0, 0x7fc0 if dt == 0 else 0x7e00)
    L_.ffcsim_set_fused_kx(S.p(xk), ctypes.c_float(0.5))
    try:
        y1 = S.sim_conv_fwd(M, dt, xb, kf1)
    finally:
        L_.ffcsim_set_fused_kx(None, ctypes.c_float(1.0))
    assert np.array_equal(kf0, kf1) and np.array_equal(y0, y1)
    # backward: slabs + ffc_kernel_ifft_grad_c against the tail's pair-plane output
    upw = L_.ffcsim_upw(M)
    ws = np.full(upw * H * nt * 2048, np.nan, np.float32)
    du0 = np.zeros_like(xb)
    nslab = L_.ffcsim_conv_bwd(M, dt, S.p(db), S.p(xb), S.p(kf0), None, None, S.p(du0), None, None, S.p(ws), B, H, M, 1)
    assert nslab > 0
    out0 = np.zeros((2, H, M), np.uint16)
    assert L_.ffcsim_kernel_ifft_grad_c(M, S.p(ws), nslab, H, S.p(out0), ctypes.c_float(0.25)) == 0
    out1 = np.full((2, H, M), 0x7fc0, np.uint16)
    du1 = np.zeros_like(xb); ws1 = np.full_like(ws, np.nan)
    L_.ffcsim_set_fused_dkpair(S.p(out1), ctypes.c_float(0.25))
    try:
        assert L_.ffcsim_conv_bwd(M, dt, S.p(db), S.p(xb), S.p(kf0), None, None, S.p(du1), None, None, S.p(ws1), B, H, M, 1) > 0
    finally:
        L_.ffcsim_set_fused_dkpair(None, ctypes.c_float(1.0))
    assert np.array_equal(du0, du1) and np.isnan(ws1).all()
    assert rel(S.from_bits(out1, 0), S.from_bits(out0, 0).astype(np.float64)) < 2e-3


@pytest.mark.parametrize("N,fac,Lk,dt", [(65536, ((16,), 4096), 40000, 0), (131072, ((32,), 4096), 131072, 1), (131072, ((32,), 4096), 70001, 0),
                                         (262144, ((64,), 4096), 100000, 1), (524288, ((128,), 4096), 131072, 0),
                                         (1048576, ((16, 16), 4096), 300004, 1)])
def test_levels_read_fp32_filter_and_write_fp32_dk(N, fac, Lk, dt):
    """BigArgs::lf32 (round 4): the first forward level reads the fp32 filter itself (prescale 2^8 in fp16 mode, one rounding to the
    16-bit type in its row load) and the last inverse level writes dk as fp32 -- bit for bit what the cast passes around the levels
    produced (ops.to_dtype_rows / to_float_rows), for the plain levels (16, 32), the R-pass levels (64, 128: one launch and one
    launch per pass), two levels, ragged lengths (element-wise accesses) and both dtypes."""
    from flashfftconv import bigfft as BG
    rng = np.random.default_rng(N + Lk)
    H = 2
    k = (rng.standard_normal((H, Lk)) * 0.05).astype(np.float32)
    new, old = S.SimOps(), S.SimOps()
    old.LONG_F32 = False
    x_new, sc_new = BG.kernel_rows(new, dt, N, k, H, Lk, fac)
    x_old, sc_old = BG.kernel_rows(old, dt, N, k, H, Lk, fac)
    assert sc_new == sc_old and np.array_equal(x_new, x_old)
    if fac[0][0] in (64, 128):
        per_pass = S.SimOps(); per_pass.one_launch = False
        assert np.array_equal(BG.kernel_rows(per_pass, dt, N, k, H, Lk, fac)[0], x_old)
    # dk side: any complex rows will do (bf16 arithmetic whatever the module dtype)
    rows = 1
    for n0 in fac[0]:
        rows *= n0
    y = S.to_bits(rng.standard_normal((2, H * rows, fac[1])).astype(np.float32), 0)
    dk_new = BG.dk_from_pair(new, N, y, H, Lk, fac)
    dk_old = BG.dk_from_pair(old, N, y, H, Lk, fac)
    assert dk_new.dtype == np.float32 and dk_new.shape == (H, Lk) and not np.isnan(dk_new).any()
    assert np.array_equal(dk_new, dk_old)
    if fac[0][0] in (64, 128):      # one launch per pass: passes c > 0 add to the stored fp32 rows (values of the 16-bit type, as before)
        pp_old = S.SimOps(); pp_old.one_launch = False; pp_old.LONG_F32 = False
        assert np.array_equal(BG.dk_from_pair(per_pass, N, y, H, Lk, fac), BG.dk_from_pair(pp_old, N, y, H, Lk, fac))


@pytest.mark.parametrize("N,fac,L,gated", [(65536, ((16,), 4096), 30000, False), (131072, ((32,), 4096), 131072, True), (262144, ((64,), 4096), 131072, False),
                                           (524288, ((128,), 4096), 100004, True)])
def test_batch_of_one_keeps_half_of_the_level_rows(N, fac, L, gated):
    """round 6 (csrc/ffc_big.h BigArgs::half): ONE real row per head on the long side (B = 1; the filter and dk always) -- the level's rows
    k0 and K - k0 are conjugate mirrors, so the levels store / read the K / 2 + 1 rows k0 <= K / 2 only and the inner kernel convolves half
    as many.  Forward (gated, ragged) and dk against the oracle, and against the full-row form of the same run (equal to rounding)."""
    from flashfftconv import bigfft as BG
    rng = np.random.default_rng(N + L)
    dt, H, B, M = 0, 2, 1, fac[1]
    K = fac[0][0]
    full, half = S.SimOps(), S.SimOps()
    half.half = True
    u, g1, g2, d = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.05).astype(np.float32)
    ub, g1b, g2b, db = (S.to_bits(x, dt) for x in (u, g1, g2, d))
    outs, dks = [], []
    for ops in (full, half):
        kf = BG.kernel_fft(ops, dt, N, k, H, L, fac)
        x = BG.levels_forward(ops, dt, N, ub, B, H, L, g1b if gated else None, fac)
        rows = K // 2 + 1 if ops.half else K
        assert x.shape == (2, H * rows, M) and kf.shape[0] == H * rows
        y = ops.conv(dt, M, x, kf, False)
        out = np.zeros_like(ub)
        BG.levels_inverse(ops, dt, N, y, out, B, H, L, g2b if gated else None, None, fac)
        outs.append(S.from_bits(out, dt))
        xd = BG.levels_forward(ops, dt, N, db, B, H, L, None, fac)
        xu = BG.levels_forward(ops, dt, N, ub, B, H, L, None, fac)
        dks.append(BG.dk_from_slabs(ops, N, ops.dkf(dt, M, xd, xu), xu.shape[0], H, L, None, fac))
    ref = O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype="bf16") if gated else O.ref_fft_conv(q(u, dt), k, N)
    _, dkref = O.ref_grads(q(u, dt), k, q(d, dt), N)
    assert rel(outs[1], ref) < 1.5e-2 and rel(dks[1], dkref) < 1.5e-2
    assert rel(outs[1], outs[0].astype(np.float64)) < 1e-2 and rel(dks[1], dks[0].astype(np.float64)) < 1e-2
    if K > 32:      # one launch per pass (ffc_outer_pass_r) == all passes in one launch (ffc_outer_pass_all): the forward rows bit for bit
        per_pass = S.SimOps(); per_pass.half = True; per_pass.one_launch = False
        xa = BG.levels_forward(half, dt, N, ub, B, H, L, None, fac)
        assert np.array_equal(xa, BG.levels_forward(per_pass, dt, N, ub, B, H, L, None, fac))
        out2 = np.zeros_like(ub)
        BG.levels_inverse(per_pass, dt, N, half.conv(dt, M, xa, BG.kernel_fft(half, dt, N, k, H, L, fac), False), out2, B, H, L, None, None, fac)
        assert rel(S.from_bits(out2, dt), O.ref_fft_conv(q(u, dt), k, N)) < 1.5e-2
