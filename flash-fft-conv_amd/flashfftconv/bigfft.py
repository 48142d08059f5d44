"""FFT sizes 65536 .. 4194304: one or two HBM-level outer DFT passes (factor 16 or 32, csrc/ffc_big.h)
around the fused kernel of size M <= 32768.  Mirrors the reference's butterfly -> inner complex
monarch -> butterfly_ifft chain (flashfftconv/conv.py:692-1732, :1867-2002, :3340-4958) but
 * every level packs two batch rows as one complex sequence (no real/complex kernel split),
 * the inner kernel is the SAME fused kernel: the complex intermediate is stored as a
   "pair-plane" tensor (2*pairs, H*prod(N0), M) whose rows 2p/2p+1 are Re/Im, which is exactly the
   (B, H, L) layout the fused kernel already consumes with "heads" = (head, k0).
The functions are written against an `ops` object (GPU: flashfftconv.conv._TorchOps, CPU tests: the
wave simulator) so the index/scale bookkeeping is verified without a GPU."""

# N -> (outer factors, fused inner size).  Inner size 65536 is the 2-pass form of the fused 32768 kernel (csrc/ffc_body.h
# struct Pass): with it fft 2M needs ONE HBM-level outer pass (factor 32) instead of two (measured, B16 H48 L=1M: forward
# 156.6 -> 141.4 ms, backward 271 -> 266 ms rescaled to H=768).  The same trick for 4M (32 x the 4-pass fft 131072) re-reads
# the full-length rows four times and loses to the two-level form (cfg4 forward 1.33 -> 1.53 ms): 4M stays on two levels.
# FFC_BIG_2LEVEL=1 / FFC_BIG_1LEVEL=1 force the two-level / one-level form of both sizes for A/B runs.
import os as _os
BIG_FACTORS = {
    65536: ((16,), 4096),
    131072: ((32,), 4096),
    262144: ((16,), 16384),
    524288: ((16,), 32768),
    1048576: ((32,), 32768),
    2097152: ((32,), 65536),
    4194304: ((16, 16), 16384),
}
if _os.environ.get("FFC_BIG_2LEVEL", "0") == "1":
    BIG_FACTORS.update({2097152: ((16, 16), 8192), 4194304: ((16, 16), 16384)})
if _os.environ.get("FFC_BIG_1LEVEL", "0") == "1":
    BIG_FACTORS.update({2097152: ((32,), 65536), 4194304: ((32,), 131072)})


# fft 4194304 in ONE level, 128 x 32768, when the long side is at most N / 4 (the HyenaDNA shape, BASELINE config 4): the
# 128-point factor runs as 4 passes of the 32-point outer kernel (ops.outer with n0 = 128 -> ffc_outer_pass_r), each reading
# the quarter-length rows and writing its 32 of the 128 rows per head -- one HBM round trip instead of two
# (reference: the 128-point butterfly, butterfly_padded_cuda_bf16.cu:302-487, conv.py:511-551).  FFC_BIG_ONE128=0 disables it.
# The same with R = 2 for fft 2097152 when L <= N / 2: 64 x 32768 (single-pass inner kernel, which also has the saved-spectra
# backward) instead of 32 x the 2-pass fft 65536.
ONE128 = {4194304: ((128,), 32768), 2097152: ((64,), 32768)} if _os.environ.get("FFC_BIG_ONE128", "1") != "0" else {}


# Round 6: the same one-level factorisations at ANY length (the level's wide form, csrc/ffc_big.h BigBody::run_wide: up to R * 32 long-side rows, an R-point
# butterfly of the row blocks in front of the pass matrices) -- fft 2M / 4M at L = N lose an HBM level resp. the 2-pass inner kernel and a quarter of the
# step's peak memory (4M, B8 H16 gated: 8.05 -> 5.91 GB), but the level kernels pay for it in instructions: every wave of a column group repeats the
# butterfly (forward) resp. the conj-twiddle product (inverse) of the rows the R waves share.  Measured (profiles/r06_ab_wide.txt): 2M +8 % / +14 %,
# 4M +17 % / +33 % (fwd / bwd) against the round-5 routing -> OPT-IN (FFC_BIG_WIDE=1): a memory option; parity-green on the simulator and the GPU.
WIDE = _os.environ.get("FFC_BIG_WIDE", "0") == "1"


def is_wide(n0, mi, Llong):
    """a level of factor 64 / 128 whose long side reaches beyond the first 32 rows"""
    return n0 in (64, 128) and Llong > 32 * mi


def choose(N, Lmax, ops=None):
    """(outer factors, fused inner size) for fft size N when no long-side row is longer than Lmax"""
    if N in ONE128 and getattr(ops, "HAS_128", False) and (Lmax <= N // (ONE128[N][0][0] // 32) or (WIDE and getattr(ops, "HAS_WIDE", False))):
        return ONE128[N]
    return BIG_FACTORS[N]


def row_freq(N, fac=None):
    """Where the inner k_f rows of one head sit in the N-point spectrum: (offsets, stride) with natural frequency
    f = offsets[row] + stride * f_inner (mod N), rows in the order the levels produce them (head-major, first level outermost).
    A level of factor n0 splits f = k0 + n0 f'; the factors 64 / 128 (R passes of the 32-point kernel) order their rows
    c * 32 + d for k0 = c + R d.  Used to mask k_f / dk_f of the frequency-sparse convolution (flashfftconv/sparse_conv.py)."""
    factors, M = fac or BIG_FACTORS[N]
    offs, stride = [0], 1
    for n0 in factors:
        if n0 in (64, 128):
            R = n0 // 32
            k0s = [c + R * d for c in range(R) for d in range(32)]
        else:
            k0s = list(range(n0))
        offs = [o + stride * k0 for o in offs for k0 in k0s]
        stride *= n0
    return offs, stride


def rows_of(ops, n0):
    """short-side rows per head of a level of factor n0: all n0 of them, or -- ops.half (round 6: ONE REAL row per head on the long side, a
    batch of one; csrc/ffc_big.h BigArgs::half) -- the n0 / 2 + 1 rows k0 <= n0 / 2, whose conjugate mirrors are never stored or convolved"""
    return n0 // 2 + 1 if getattr(ops, "half", False) else n0


def half_ok(N, B, Lmax, ops=None):
    """the half-row form applies to a batch of ONE real row per head through a SINGLE level (the mirror relation is per level)"""
    factors, M = choose(N, Lmax, ops)
    return B == 1 and len(factors) == 1 and not is_wide(factors[0], N // factors[0], Lmax)      # (the wide form stores all rows)


def level_scale(n0):
    """forward scale of one level ~ 1/sqrt(N0) (keeps the spectrum RMS near the input RMS)"""
    return {16: 0.25, 32: 0.125, 64: 0.125, 128: 0.0625}[n0]


def inner_sfwd(M):
    """the fused plan's forward scale s_fwd = 2^-ceil(log2(M) / 2) (csrc/ffc_plan.cpp)"""
    lg = M.bit_length() - 1
    return 2.0 ** (-((lg + 1) // 2))


def levels_forward(ops, dt, N, x, B_valid, H, L, gate=None, fac=None, lf32=None):
    """x: (B_valid, H, L) long-side tensor -> (2*npair, H*prod(N0), M) pair-plane tensor.  fac: choose(N, ...) (default: BIG_FACTORS).
    lf32: x is fp32 (the filter k); the first level multiplies by this power of two and rounds to dt itself (ops.LONG_F32)."""
    factors, M = fac or BIG_FACTORS[N]
    npair = (B_valid + 1) // 2
    Hx, nlev, Llong, bv = H, N, L, B_valid
    for i, n0 in enumerate(factors):
        mi = nlev // n0
        out = ops.empty_pair(dt, 2 * npair, Hx * rows_of(ops, n0), mi)
        if i == 0 and lf32 is not None:
            ops.outer(dt, n0, True, x, out, gate, bv, npair, Hx, mi, Llong, level_scale(n0), lf32=lf32)
        else:
            ops.outer(dt, n0, True, x, out, gate if i == 0 else None, bv, npair, Hx, mi, Llong, level_scale(n0))
        x, Hx, nlev, Llong, bv = out, Hx * rows_of(ops, n0), mi, mi, 2 * npair
    return x


def levels_inverse(ops, dt, N, y, out, B_valid, H, L, gate=None, shared=None, fac=None, lf32=False):
    """y: (2*npair, H*prod(N0), M) -> out (B_valid, H, L) (written in place).  `shared` caches the
    intermediate of the two-level case so several gated outputs reuse it.  lf32: `out` is fp32 (dk), written by the last level."""
    factors, M = fac or BIG_FACTORS[N]
    npair = y.shape[0] // 2
    Hx = H
    for n0 in factors:
        Hx *= rows_of(ops, n0)
    nlev = M
    cur = y
    for i in reversed(range(len(factors))):
        n0 = factors[i]
        Hx //= rows_of(ops, n0)
        nlev *= n0
        sc = 1.0 / (n0 * level_scale(n0))
        if i == 0 and lf32:
            ops.outer(dt, n0, False, cur, out, gate, B_valid, npair, Hx, nlev // n0, L, sc, lf32=1.0)
        elif i == 0:
            ops.outer(dt, n0, False, cur, out, gate, B_valid, npair, Hx, nlev // n0, L, sc)
        else:
            if shared is not None and "mid" in shared:
                cur = shared["mid"]
            else:
                mid = ops.empty_pair(dt, 2 * npair, Hx, nlev)
                ops.outer(dt, n0, False, cur, mid, None, 2 * npair, npair, Hx, nlev // n0, nlev, sc)
                cur = mid
                if shared is not None:
                    shared["mid"] = mid
    return out


def prod_scale(N, fac=None):
    factors, M = fac or BIG_FACTORS[N]
    s = 1.0
    for n0 in factors:
        s *= level_scale(n0)
    return s


def _k_levels(ops, dt, N, k, H, Lk, pre, fac):
    """the levels over the filter k (H, Lk) fp32: the first level reads the fp32 rows itself (prescale and rounding to dt in its row
    load, round 4) where the backend can (ops.LONG_F32); otherwise a cast pass in front (ops.to_dtype_rows)"""
    if getattr(ops, "LONG_F32", False):
        return levels_forward(ops, dt, N, ops.f32_rows(k, H, Lk), 1, H, Lk, None, fac, lf32=pre)
    return levels_forward(ops, dt, N, ops.to_dtype_rows(dt, k, H, Lk, pre), 1, H, Lk, None, fac)


def _dk_levels(ops, N, y, H, Lk, fac):
    """complex rows (2, H*prod(N0), M) bf16 -> dk (H, Lk) fp32 through the inverse levels; the last one writes fp32 itself (ops.LONG_F32)"""
    BF = ops.BF16
    if getattr(ops, "LONG_F32", False):
        out = ops.empty_f32(1, H, Lk)
        levels_inverse(ops, BF, N, y, out, 1, H, Lk, None, None, fac, lf32=True)
        return out[0]
    out = ops.empty_pair(BF, 1, H, Lk)
    levels_inverse(ops, BF, N, y, out, 1, H, Lk, None, None, fac)
    return ops.to_float_rows(out, H, Lk)


def kernel_fft(ops, dt, N, k, H, Lk, fac=None):
    """k (H, Lk) fp32 -> inner k_f rows (H*prod(N0), M-internal), unscaled K_f."""
    factors, M = fac or BIG_FACTORS[N]
    pre = ops.k_prescale(dt)                         # 2^8 in fp16 mode (k's energy sits in a few taps)
    x = _k_levels(ops, dt, N, k, H, Lk, pre, fac)
    hp = x.shape[1]
    return ops.kfft_c(dt, M, x, hp, 1.0 / (inner_sfwd(M) * prod_scale(N, fac) * pre))


def kernel_rows(ops, dt, N, k, H, Lk, fac=None):
    """k (H, Lk) fp32 -> (x, scale): the complex inner rows (2, H*prod(N0), M) whose M-point transform, times `scale`, is the
    unscaled k_f of kernel_fft -- for callers that run that last transform inside their convolution launch (ffc_conv_fwd_kx)"""
    factors, M = fac or BIG_FACTORS[N]
    pre = ops.k_prescale(dt)
    return _k_levels(ops, dt, N, k, H, Lk, pre, fac), 1.0 / (inner_sfwd(M) * prod_scale(N, fac) * pre)


def dk_pair_scale(N, fac=None):
    """scale of the inner dk_f -> complex rows step (ffc_kernel_ifft_grad_c / the backward launch's tail)"""
    factors, M = fac or BIG_FACTORS[N]
    return 1.0 / (inner_sfwd(M) * prod_scale(N, fac))


def dk_from_pair(ops, N, y, H, Lk, fac=None):
    """complex rows of the inverted inner dk_f, (2, H*prod(N0), M) bf16 -> dk (H, Lk) fp32: the levels of dk_from_slabs"""
    return _dk_levels(ops, N, y, H, Lk, fac)


def dk_from_slabs(ops, N, ws, Bp, H, Lk, nslab=None, fac=None):
    """fp32 W slabs of the inner size -> dk (H, Lk) fp32.  Always bf16 arithmetic (fp32 range).
    nslab: `ws` holds that many caller-owned slabs (hp, kf_elems, 2) instead of a backward launch's workspace."""
    factors, M = fac or BIG_FACTORS[N]
    hp = H
    for n0 in factors:
        hp *= rows_of(ops, n0)
    BF = ops.BF16
    sc = 1.0 / (inner_sfwd(M) * prod_scale(N, fac))
    y = ops.dkifft_c(M, ws, Bp, hp, sc) if nslab is None else ops.dkifft_c(M, ws, Bp, hp, sc, nslab)     # (2, hp, M) bf16
    return _dk_levels(ops, N, y, H, Lk, fac)
