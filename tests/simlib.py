"""ctypes access to the CPU wave simulator (libffcsim.so) + host-side helpers shared by tests."""
import ctypes, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flash-fft-conv_amd")
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
DT_BF16, DT_F16 = 0, 1


def build_sim(force=False):
    so = os.path.join(LIBDIR, "libffcsim.so")
    srcs = [os.path.join(CSRC, f) for f in ("ffc_sim.cpp", "ffc_plan.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("ffc_body.h", "ffc_modes.h", "ffc_layout.h", "ffc_plan.h", "ffc_big.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        os.makedirs(LIBDIR, exist_ok=True)
        subprocess.check_call(["g++", "-O0",   # one large translation unit: -O1 compiles 7.5 min for a 23 s test run, -O0 50 s for 40 s
                                "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so] + srcs)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.environ.get("FFC_SIM_LIB") or build_sim())      # FFC_SIM_LIB: a hand-built variant (e.g. -DFFC_DYN_TILES=1)
    return _lib


def f32_to_bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def to_bits(x, dtype):
    return f32_to_bf16_bits(x) if dtype == DT_BF16 else np.ascontiguousarray(np.asarray(x, np.float32).astype(np.float16)).view(np.uint16)


def from_bits(b, dtype):
    b = np.ascontiguousarray(b)
    return bf16_bits_to_f32(b) if dtype == DT_BF16 else b.view(np.float16).astype(np.float32)


def plan_info(N, dtype):
    nt = ctypes.c_int(); sf = ctypes.c_double(); sk = ctypes.c_double()
    rc = lib().ffcsim_plan_info(N, dtype, ctypes.byref(nt), ctypes.byref(sf), ctypes.byref(sk), None)
    assert rc == 0, f"unsupported N={N}"
    freq = np.zeros(nt.value * 1024, np.int32)
    lib().ffcsim_plan_info(N, dtype, ctypes.byref(nt), ctypes.byref(sf), ctypes.byref(sk),
                           freq.ctypes.data_as(ctypes.c_void_p))
    return nt.value, sf.value, sk.value, freq


def make_kf_internal(k, N, dtype):
    """k (H, Lk) float -> internal-order k_f bits (H, NT*1024, 2), via float64 numpy FFT."""
    nt, sf, sk, freq = plan_info(N, dtype)
    kf = np.fft.fft(np.asarray(k, np.float64), n=N, axis=-1)[:, freq] * sk
    out = np.stack([kf.real, kf.imag], -1).astype(np.float32)
    return to_bits(out, dtype)


def make_kf_from_spectrum(kf_natural, N, dtype):
    """natural-order complex spectrum (H, N) -> internal-order k_f bits (H, NT*1024, 2)"""
    nt, sf, sk, freq = plan_info(N, dtype)
    kf = np.asarray(kf_natural)[:, freq] * sk
    return to_bits(np.stack([kf.real, kf.imag], -1).astype(np.float32), dtype)


def p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def sim_conv_fwd(N, dtype, u_bits, kf_bits, pre=None, post=None, conj=0):
    B, H, L = u_bits.shape
    y = np.zeros_like(u_bits)
    rc = lib().ffcsim_conv_fwd(N, dtype, p(u_bits), p(kf_bits), p(pre), p(post), p(y), B, H, L, conj)
    assert rc == 0, rc
    return y


def sim_kernel_fft(N, dtype, k):
    k = np.ascontiguousarray(k, np.float32)
    H, Lk = k.shape
    nt, _, _, _ = plan_info(N, dtype)
    kf = np.zeros((H, nt * 1024, 2), np.uint16)
    rc = lib().ffcsim_kernel_fft(N, dtype, p(k), H, Lk, p(kf))
    assert rc == 0, rc
    return kf


def sim_dk(N, dtype, dout_bits, u_bits, Lk, pre=None, post=None, nchunk=1):
    """dk (H, Lk) fp32 through the simulated dkf + dkifft kernels."""
    B, H, L = u_bits.shape
    nt, _, _, _ = plan_info(N, dtype)
    upw = lib().ffcsim_upw(N)
    ws = np.full(max(nchunk, 1) * upw * H * nt * 2048, np.nan, np.float32)
    nslab = lib().ffcsim_conv_bwd_dkf(N, dtype, p(dout_bits), p(u_bits), p(pre), p(post), p(ws), B, H, L, nchunk)
    assert nslab > 0, nslab
    dk = np.full((H, Lk), np.nan, np.float32)
    rc = lib().ffcsim_kernel_ifft_grad(N, dtype, p(ws), nslab, H, Lk, p(dk))
    assert rc == 0, rc
    return dk


# ---------------------------------------------------------------- big FFT sizes on the simulator
class SimOps:
    """`ops` backend of flashfftconv.bigfft for the CPU wave simulator (numpy uint16 bit tensors)."""
    BF16 = DT_BF16

    def __init__(self):
        self.nslab = None

    def empty_pair(self, dt, Bp, Hx, n):
        return np.zeros((Bp, Hx, n), np.uint16)

    HAS_128 = True
    HAS_WIDE = True
    one_launch = True

    LONG_F32 = True      # the levels read the fp32 filter / write the fp32 dk themselves (set False: cast passes around them)

    def f32_rows(self, k, H, Lk):
        return np.ascontiguousarray(np.asarray(k, np.float32).reshape(1, H, Lk))

    def empty_f32(self, Bp, Hx, n):
        return np.full((Bp, Hx, n), np.nan, np.float32)

    half = False         # one REAL row per head on the long side: the levels keep the rows k0 <= K / 2 only (csrc/ffc_big.h BigArgs::half)

    def outer(self, dt, n0, fwd, inp, out, gate, bv, npair, Hin, mi, Llong, scale, lf32=None):
        if self.half:
            assert bv == 1 and npair == 1
            dt = dt | 32
        if lf32 is not None:       # the library's dtype flags (csrc/ffc_k_big.hip decode_dtype)
            e = int(round(np.log2(lf32)))
            assert 2.0 ** e == lf32 and (inp if fwd else out).dtype == np.float32
            dt = dt | 16 | (e << 8)
        if n0 in (64, 128) and Llong > 32 * mi:      # the wide form (ffc_outer_pass_all beyond the first 32 long-side rows): 16-bit rows, see _TorchOps.outer
            base = dt & 15
            assert not self.half
            if lf32 is not None and fwd:
                inp = to_bits(np.asarray(inp, np.float32) * np.float32(lf32), base)
            tmp = np.zeros(out.shape, np.uint16) if (lf32 is not None and not fwd) else out
            rc = lib().ffcsim_big_outer_all(n0 // 32, base, int(fwd), p(inp), p(tmp), p(gate), bv, npair, Hin, mi, Llong, ctypes.c_float(scale))
            assert rc == 0, rc
            if tmp is not out:
                out[...] = from_bits(tmp, base)
            return
        if n0 in (64, 128) and self.one_launch:      # all R passes in one workgroup run (ffc_outer_pass_all)
            rc = lib().ffcsim_big_outer_all(n0 // 32, dt, int(fwd), p(inp), p(out), p(gate), bv, npair, Hin, mi, Llong, ctypes.c_float(scale))
            assert rc == 0, rc
            return
        if n0 in (64, 128):        # factor R * 32 = R passes of the 32-point kernel (ffc_outer_pass_r)
            for c in range(n0 // 32):
                rc = lib().ffcsim_big_outer_r(32, n0 // 32, c, dt, int(fwd), p(inp), p(out), p(gate), bv, npair, Hin, mi, Llong,
                                              ctypes.c_float(scale))
                assert rc == 0, rc
            return
        rc = lib().ffcsim_big_outer(n0, dt, int(fwd), p(inp), p(out), p(gate), bv, npair, Hin, mi, Llong,
                                    ctypes.c_float(scale))
        assert rc == 0, rc

    def k_prescale(self, dt):
        return 256.0 if dt == DT_F16 else 1.0

    def to_dtype_rows(self, dt, k, H, Lk, pre=1.0):
        return to_bits(np.asarray(k, np.float32).reshape(1, H, Lk) * np.float32(pre), dt)

    def to_float_rows(self, out, H, Lk):
        return from_bits(out, DT_BF16)[0].astype(np.float32)

    def kfft_c(self, dt, M, x, hp, scale):
        nt, _, _, _ = plan_info(M, dt)
        kf = np.zeros((hp, nt * 1024, 2), np.uint16)
        rc = lib().ffcsim_kernel_fft_c(M, dt, p(x), hp, p(kf), ctypes.c_float(scale))
        assert rc == 0, rc
        return kf

    def conv(self, dt, M, x, kf, conj):
        return sim_conv_fwd(M, dt, x, kf, conj=int(conj))

    def dkf(self, dt, M, xd, xu):
        Bp, hp, _ = xu.shape
        nt, _, _, _ = plan_info(M, dt)
        upw = lib().ffcsim_upw(M)
        ws = np.full(upw * hp * nt * 2048, np.nan, np.float32)
        self.nslab = lib().ffcsim_conv_bwd_dkf(M, dt, p(xd), p(xu), None, None, p(ws), Bp, hp, M, 1)
        assert self.nslab > 0
        return ws

    def dkifft_c(self, M, ws, Bp, hp, scale):
        out = np.zeros((2, hp, M), np.uint16)
        rc = lib().ffcsim_kernel_ifft_grad_c(M, p(ws), self.nslab, hp, p(out), ctypes.c_float(scale))
        assert rc == 0, rc
        return out


def sim_bwd(N, dtype, dout_bits, u_bits, kf_bits, Lk, pre=None, post=None, nchunk=1, fused_dk=False):
    """Fused backward on the simulator: returns (du bits, dpre bits or None, dk fp32).  fused_dk: dk written by the backward
    kernel itself from its accumulation registers (DkfArgs::dk_out; fft 4096 .. 32768 bf16, one chunk) instead of slabs + dkifft."""
    if fused_dk:
        B, H, L = u_bits.shape
        nt, _, _, _ = plan_info(N, dtype)
        ws = np.full(lib().ffcsim_upw(N) * H * nt * 2048, np.nan, np.float32)
        du = np.zeros_like(u_bits)
        dpre = np.zeros_like(u_bits) if pre is not None else None
        dpost = np.zeros_like(u_bits) if pre is not None else None
        dk = np.full((H, Lk), np.nan, np.float32)
        lib().ffcsim_set_fused_dk(p(dk), Lk)
        try:
            assert lib().ffcsim_conv_bwd(N, dtype, p(dout_bits), p(u_bits), p(kf_bits), p(pre), p(post), p(du), p(dpre), p(dpost),
                                         p(ws), B, H, L, 1) > 0
        finally:
            lib().ffcsim_set_fused_dk(None, 0)
        assert np.isnan(ws).all(), "the fused dk tail must not write slabs"
        return du, dpre, dk
    B, H, L = u_bits.shape
    nt, _, _, _ = plan_info(N, dtype)
    upw = lib().ffcsim_upw(N)
    ws = np.full(max(nchunk, 1) * upw * H * nt * 2048, np.nan, np.float32)
    du = np.zeros_like(u_bits)
    dpre = np.zeros_like(u_bits) if pre is not None else None
    dpost = np.zeros_like(u_bits) if pre is not None else None
    nslab = lib().ffcsim_conv_bwd(N, dtype, p(dout_bits), p(u_bits), p(kf_bits), p(pre), p(post), p(du), p(dpre), p(dpost),
                                  p(ws), B, H, L, nchunk)
    assert nslab > 0, nslab
    dk = np.full((H, Lk), np.nan, np.float32)
    assert lib().ffcsim_kernel_ifft_grad(N, dtype, p(ws), nslab, H, Lk, p(dk)) == 0
    sim_bwd.dpost = dpost          # fused sizes >= 4096 only (N <= 1024 leaves zeros: the library runs the forward kernel)
    return du, dpre, dk


def sim_fwd_bwd_z(N, dtype, u_bits, dout_bits, kf_bits, pre=None, post=None, nchunk=1, flags=0, keep="zy"):
    """the spectrum-saving pair on the simulator (ffc_conv_fwd_z -> ffc_conv_bwd_z / _zy): forward output, du, dpre, dpost bits and
    the fp32 dk_f slabs.  flags: ConvArgs::flags of the backward (8 = no LDS-DMA input rows).  keep = "y": the gated single-tile form that
    keeps the output before the postgate WITHOUT the spectra (fft <= 2048; the backward transforms u * pregate again)."""
    B, H, L = u_bits.shape
    nt, _, _, _ = plan_info(N, dtype)
    upw = lib().ffcsim_upw(N)
    # (single-tile sizes: one 4 KB slot per tile of G pairs and pass, ffc_spectrum_bytes -- a slot per pair is enough room)
    z = np.zeros(((B + 1) // 2) * H * max(N, 1024) * 2, np.uint16) if "z" in keep else None
    yraw = np.zeros_like(u_bits) if pre is not None else None
    assert z is not None or (yraw is not None and N <= 2048)
    y = np.zeros_like(u_bits)
    lib().ffcsim_set_z(p(z), p(yraw), 0)
    try:
        assert lib().ffcsim_conv_fwd(N, dtype, p(u_bits), p(kf_bits), p(pre), p(post), p(y), B, H, L, 0) == 0
        ws = np.full(max(nchunk, 1) * upw * H * nt * 2048, np.nan, np.float32)
        du = np.zeros_like(u_bits)
        dpre = np.zeros_like(u_bits) if pre is not None else None
        dpost = np.zeros_like(u_bits) if pre is not None else None
        lib().ffcsim_set_z(p(z), p(yraw), flags)
        nslab = lib().ffcsim_conv_bwd(N, dtype, p(dout_bits), p(u_bits), p(kf_bits), p(pre), p(post), p(du), p(dpre), p(dpost),
                                      p(ws), B, H, L, nchunk)
        assert nslab > 0, nslab
    finally:
        lib().ffcsim_set_z(None, None, 0)
    return y, du, dpre, dpost, ws[: nslab * H * nt * 2048].copy()
