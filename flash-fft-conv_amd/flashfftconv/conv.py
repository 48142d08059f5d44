"""FlashFFTConv module + autograd for MI355X (drop-in for reference flashfftconv/conv.py).

    conv = FlashFFTConv(seqlen, dtype=torch.bfloat16).to(device)
    y = conv(u, k)                        # u (B,H,L) bf16/fp16, k (H,Lk) fp32  ->  (B,H,L)
    y = conv(u, k, pregate, postgate)     # y = postgate * conv(u * pregate, k)

Semantics: y = iFFT(FFT(u, N) * FFT(k, N)).real[..., :L] with N = seqlen (reference
tests/test_flashfftconv.py:5-13).  The reference's 4.9 KLoC `if seqlen == ...` dispatch
(conv.py:563-4958) collapses to one table-driven plan inside the HIP library."""
import ctypes
import math
import re
import threading
import torch

from . import _lib
from . import bigfft as _big

_DT = {torch.bfloat16: 0, torch.float16: 1}
FUSED_SEQLENS = (256, 512, 1024, 4096, 8192, 16384, 32768)
# fft sizes that run as R passes of a fused kernel (csrc/ffc_body.h struct Pass) instead of an HBM-level outer pass around a
# smaller fused kernel (flashfftconv/bigfft.py): 2048 (2 passes of the 1024 kernel), 65536 and 131072 (2 / 4 passes of the
# 32768 kernel) -- all three by default.  Measured: 65536 1.4x faster than the HBM level (profiles/r02_multipass.txt); 131072
# forward 1.12x, and with the spectra kept for the backward pass (round 3) its fwd+bwd 15.8 -> 12.8 ms (profiles/r03_spectrum.txt)
# where the HBM-level form measured 16.5.  FFC_MULTIPASS="2048,65536" etc. selects other routings for A/B runs.
import os as _os
MULTIPASS_SEQLENS = tuple(int(x) for x in _os.environ.get("FFC_MULTIPASS", "2048,65536,131072").split(",") if x.strip())
# FlashFFTConv._fit_seqlen: run the smallest fft size that still holds the linear convolution of the rows handed in (fft sizes > 32768)
_FIT_FFT = _os.environ.get("FFC_FIT_FFT", "1") != "0"
_ROUTE_BY_LENGTH = _os.environ.get("FFC_ROUTE_BY_LENGTH", "1") != "0"      # fft 131072: HBM-level form for rows longer than N/2 (FlashFFTConv._route_big)
# fft size 2048 has no 16/32-digit factorisation of its own.  By default it runs as 2 passes of the 1024 kernel
# (MULTIPASS_SEQLENS below); the round-1 form, kept for A/B runs (FFC_MULTIPASS without 2048): the 4096 plan with k periodised,
# k' = [k_2048 | k_2048].  FFT_4096(k') is 2*FFT_2048(k) on the even bins and 0 on the odd ones, so the 4096-point
# circular convolution with k' IS the 2048-point circular convolution with k (u occupies <= 2048 samples, the
# L <= N/2 kernel variant).  dk folds back the same way: dk = dk'[:2048] + dk'[2048:].
FOLDED_SEQLENS = {2048: 4096}
SUPPORTED_SEQLENS = tuple(sorted(FUSED_SEQLENS + tuple(FOLDED_SEQLENS))) + tuple(sorted(_big.BIG_FACTORS))


def _periodise_k(k, n):
    """(H, Lk <= n) -> (H, 2n) fp32: k zero-padded to n and repeated twice."""
    k = torch.nn.functional.pad(k.detach().to(torch.float32), (0, n - k.shape[-1]))
    return torch.cat((k, k), dim=-1)


# autograd.Function.forward always runs with grad mode off, and ctx.needs_input_grad mirrors the inputs' requires_grad flags
# whether or not a graph is being recorded: the caller's grad mode is noted around .apply so that a forward under
# torch.no_grad() does not store spectra nobody will read
_TLS = threading.local()


def _apply_noting_grad_mode(fn, *args):
    prev = getattr(_TLS, "grad", None)
    _TLS.grad = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _TLS.grad = prev


def _recording():
    g = getattr(_TLS, "grad", None)
    return True if g is None else g


_NOOP_CTX = __import__("contextlib").nullcontext()


def _dev_ctx(device):
    """device guard that costs nothing when `device` already is the current one (torch.cuda.device() resolves the index through
    several Python layers, ~5 us per entry: a tenth of a short-sequence step)"""
    get = getattr(torch._C, "_cuda_getDevice", None)
    if get is not None and device.index is not None and device.index == get():
        return _NOOP_CTX
    return torch.cuda.device(device)


def _ws_bytes(plan, B, H):
    """ffc_dkf_workspace_bytes, remembered per (B, H): one ctypes trip less per backward"""
    c = plan.__dict__.setdefault("_ws_cache", {})
    n = c.get((B, H))
    if n is None:
        n = c[(B, H)] = _lib.lib().ffc_dkf_workspace_bytes(plan.handle, B, H)
    return n


def _kf_key(k):
    return (k.data_ptr(), k._version, tuple(k.shape), k.dtype, k.device)


class _Plan:
    """Owns an ffc_plan (DFT tiles + twiddles on the device) for one (fft size, dtype, device)."""

    def __init__(self, seqlen, dtype, device):
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().ffc_plan_create(seqlen, _DT[dtype], ctypes.byref(self.handle)), "ffc_plan_create")
        self.kf_elems = _lib.lib().ffc_plan_kf_elems(self.handle)
        self.seqlen, self.dtype, self.device = seqlen, dtype, device

    def __del__(self):
        try:
            if self.handle:
                with torch.cuda.device(self.device):     # hipFree of the tables on the device that owns them
                    _lib.lib().ffc_plan_destroy(self.handle)
        except Exception:
            pass


# Plans are process-wide, keyed by (fft size, dtype, device index): modules hold no ctypes state, so they deep-copy and
# pickle like the reference module (EMA / SWA copies, torch.save(model)); two modules of one size share the tables.
_PLANS = {}
_PLANS_LOCK = threading.Lock()


def get_plan(seqlen, dtype, device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (seqlen, dtype, idx)
    p = _PLANS.get(key)
    if p is None:
        with _PLANS_LOCK:
            p = _PLANS.get(key)
            if p is None:
                p = _Plan(seqlen, dtype, torch.device("cuda", idx))
                _PLANS[key] = p
    return p


def reload_env():
    """tuning scripts: re-read FFC_* environment knobs into every cached plan (launches never read the environment)"""
    for p in list(_PLANS.values()):
        _lib.lib().ffc_plan_reload_env(p.handle)
        p.__dict__.pop("_ws_cache", None)      # the workspace size depends on FFC_WG_MULT (chunks per head): ADVICE r04


# Buffers the REFERENCE module registers (persistent, so they are in every checkpoint saved from it):
# flashfftconv/conv.py:89-92, 106-109, ... (f_32_fft, f_sqrt_N_ifft, twiddle_factors_fft_32_1K, f_128_fft_real, twid, ...)
_REF_BUFFER = re.compile(r"^(f_(\d+|sqrt_N)_i?fft(_real|_imag)?|twiddle_factors_i?fft(_\w+)?|twid)$")


def _conv(plan, u, kf, pregate, postgate, conj):
    B, H, L = u.shape
    y = torch.empty_like(u)
    _lib.check(_lib.lib().ffc_conv_fwd(plan.handle, _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pregate), _lib.ptr(postgate),
                                       _lib.ptr(y), B, H, L, int(conj), _lib.stream_ptr()), "ffc_conv_fwd")
    return y


def _sparse_rows(mod, plan):
    """frequency-sparse module on a plan with the compute-skipping kernel (fft 16384 / 32768): number of kept spectrum rows
    per side, k3 < rows or k3 >= 32 - rows covers every kept bin |f| < keep; 0 = dense kernel on the masked k_f."""
    if mod._kf_keep is None or plan.seqlen not in (16384, 32768) or mod._folded:
        return 0
    rows = -(-int(mod._kf_keep) // (plan.seqlen // 32))
    return rows if 1 <= rows <= 4 else 0


def _conv_sparse(plan, u, kf, pregate, postgate, conj, rows):
    B, H, L = u.shape
    y = torch.empty_like(u)
    _lib.check(_lib.lib().ffc_conv_fwd_sparse(plan.handle, _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pregate), _lib.ptr(postgate),
                                              _lib.ptr(y), B, H, L, int(conj), rows, _lib.stream_ptr()), "ffc_conv_fwd_sparse")
    return y


# Memory budget of the kept spectra: save_spectrum spends memory the reference does not (2x the bytes of u per
# layer at L = N/2, + 1x gated), and the OutOfMemoryError fallbacks below only see a failure of THESE allocations -- a model that
# fitted before could run out later, in somebody else's allocation.  So a buffer is only taken when it is at most
# FFC_SPECTRUM_FRACTION (default 1/8) of the memory that is free at that moment (device free + the caching allocator's unused
# reserve): layer after layer the rule limits itself, the total can never exceed the free memory at the first layer and the
# last 7/8 of whatever is left always stay available to the rest of the model.  module.save_spectrum = "always" skips the test.
# gated forward at the single-tile sizes: largest fft size that keeps y_raw WITHOUT the spectra (the C-ABI takes it up to 2048).  A MEMORY option, off
# by default (0: spectra and y_raw): at fft 1024 B64 H768 the kept bytes drop by a third (318 -> 217 MB peak), the forward gains 9 %, the backward --
# which then reads u and pregate twice, for the transform and for the gate gradients -- loses 10 %; fwd + bwd +2 % (profiles/r06_ab_y_only.txt)
_Y_ONLY_MAX = int(_os.environ.get("FFC_Y_ONLY_MAX", "0"))
_SPEC_FRACTION = float(_os.environ.get("FFC_SPECTRUM_FRACTION", "0.125"))
_SPEC_SMALL = 8 << 20          # below this a request is not worth a hipMemGetInfo call (the OOM fallback still covers it)
_free_cache = {}


def _free_bytes(device):
    """free device memory + unused reserve of torch's caching allocator; hipMemGetInfo is queried at most every 10 ms per device
    (the requests granted in between are subtracted from the cached figure)"""
    import time
    idx = device.index if device.index is not None else torch.cuda.current_device()
    now = time.monotonic()
    c = _free_cache.get(idx)
    if c is None or now - c[1] > 0.010:
        free, _ = torch.cuda.mem_get_info(idx)
        free += torch.cuda.memory_reserved(idx) - torch.cuda.memory_allocated(idx)
        c = _free_cache[idx] = [free, now]
    return c


def _capturing():
    try:
        return torch.cuda.is_current_stream_capturing()
    except Exception:          # no device (host-logic tests)
        return False


def _spectrum_budget_ok(nbytes, device, mode=True):
    if mode == "always" or nbytes <= _SPEC_SMALL:
        return True
    if _capturing():      # graph capture (flashfftconv/graphs.py): no hipMemGetInfo under capture; the graph's private pool decides,
        return True       # and an OutOfMemoryError still falls back
    c = _free_bytes(device)
    if nbytes > _SPEC_FRACTION * c[0]:
        return False
    c[0] -= nbytes
    return True


# how often a training forward wanted to keep spectra and took the recomputing path instead (budget or allocation failure):
# benchmarks and tests assert on it so that a silent switch of kernels cannot hide in a timing
SPECTRUM_FALLBACKS = {"budget": 0, "oom": 0}


def _spectrum_buffer(plan, B, H, device, gated=True, mode=True):
    """Buffer for the spectra FFT(u * pregate) that the forward pass keeps for the backward pass (ffc_conv_fwd_z / ffc_conv_bwd_z),
    or None: no memory for it (the caller then takes the recomputing path, like the reference).  Every fused plan has the path:
    [H][pair][fft size] complex values, for the single-tile sizes (fft <= 2048) one 4 KB slot per tile and pass."""
    # single-tile sizes without gates: not worth it (measured at B64 H768, fwd + bwd: fft 256 +5 %, 1024 +-0; gated -9 % / -17 %
    # because the saved pre-postgate output also replaces the extra forward launch that produces dpostgate there; 2048 -8 % / -13 %)
    if plan.seqlen <= 1024 and not gated:
        return None
    n = _lib.lib().ffc_spectrum_bytes(plan.handle, B, H)
    if n <= 0:
        return None
    if not _spectrum_budget_ok(n, device, mode):
        SPECTRUM_FALLBACKS["budget"] += 1
        return None
    try:
        return torch.empty(n, dtype=torch.uint8, device=device)
    except torch.cuda.OutOfMemoryError:
        SPECTRUM_FALLBACKS["oom"] += 1
        return None


def _conv_save(plan, u, kf, pregate, postgate, z, yraw=None):
    B, H, L = u.shape
    y = torch.empty_like(u)
    _lib.check(_lib.lib().ffc_conv_fwd_z(plan.handle, _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pregate), _lib.ptr(postgate), _lib.ptr(y),
                                         _lib.ptr(z), _lib.ptr(yraw), B, H, L, 0, 0, 0, 0, _lib.stream_ptr()), "ffc_conv_fwd_z")
    return y


def _kernel_fft(plan, k):
    """k (H, Lk) fp32 -> k_f in the plan's internal order (H, kf_elems, 2), plan dtype."""
    H = k.shape[0]
    kf = torch.empty(H, plan.kf_elems, 2, dtype=plan.dtype, device=k.device)
    k32 = k.detach().to(torch.float32).contiguous()
    _lib.check(_lib.lib().ffc_kernel_fft(plan.handle, _lib.ptr(k32), H, k32.shape[-1], _lib.ptr(kf), _lib.stream_ptr()),
               "ffc_kernel_fft")
    return kf


_BIG_ONE_CALL = _os.environ.get("FFC_BIG_ONE_CALL", "1") != "0"      # A/B switch: "0" = kfft_c / conv / bwd / dkifft_c as separate calls
_ONE_LAUNCH_LEVEL = _os.environ.get("FFC_BIG_ONE_LAUNCH", "1") != "0"      # A/B switch: "0" = one launch per pass (ffc_outer_pass_r)


class _TorchOps:
    """GPU backend of flashfftconv.bigfft (FFT sizes >= 65536): thin wrappers over the C-ABI."""
    BF16 = torch.bfloat16

    def __init__(self, mod, device, half=False):
        # half (round 6): the long side is ONE REAL row per head (a batch of one) -- the levels keep the rows k0 <= K / 2 only and the inner
        # kernel convolves half as many (csrc/ffc_big.h BigArgs::half, bigfft.rows_of)
        self.mod, self.device, self.half = mod, device, half

    def _plan(self, N):
        return self.mod._get_plan(self.device, N)

    def empty_pair(self, dt, Bp, Hx, n):
        return torch.empty(Bp, Hx, n, dtype=dt, device=self.device)

    HAS_128 = True      # factor 128 as 4 passes of the 32-point kernel (bigfft.choose)
    HAS_WIDE = True     # ... at any length (round 6: the level's wide form, ffc_outer_pass_all with Llong > 32 * Mi)

    # the levels read the fp32 filter / write the fp32 dk themselves (no cast kernels around them); FFC_BIG_LONG_F32=0: A/B switch
    LONG_F32 = _os.environ.get("FFC_BIG_LONG_F32", "1") != "0"

    def f32_rows(self, k, H, Lk):
        return k.detach().reshape(1, H, Lk).to(torch.float32).contiguous()

    def empty_f32(self, Bp, Hx, n):
        return torch.empty(Bp, Hx, n, dtype=torch.float32, device=self.device)

    def outer(self, dt, n0, fwd, inp, out, gate, bv, npair, Hin, mi, Llong, scale, lf32=None):
        dcode = _DT[dt]
        if self.half:              # | 32: half rows (include/flashfftconv_hip.h, the level entry points' `dtype`)
            assert bv == 1 and npair == 1
            dcode |= 32
        if lf32 is not None:       # fp32 long side: | 16, forward prescale 2^e in bits 8..15 (csrc/ffc_k_big.hip decode_dtype)
            e = int(round(math.log2(lf32)))
            assert 2.0 ** e == lf32 and (fwd or e == 0)
            assert (inp if fwd else out).dtype == torch.float32
            dcode |= 16 | (e << 8)
        if n0 in (64, 128):
            R = n0 // 32
            pr = self._plan(32768 * R)      # the R-pass plan: its per-pass outer-digit tables are the matrices of the passes
            if _big.is_wide(n0, mi, Llong):
                # the wide form moves 16-bit rows: the fp32 filter is rounded in front of it (the product with its prescale, rounded once -- what the
                # level's own fp32 load does) and the fp32 dk widened behind it (the level rounds its results to 16 bits before they leave LDS either way)
                assert not self.half
                if lf32 is not None and fwd:
                    inp = torch.mul(inp, lf32).to(dt) if lf32 != 1.0 else inp.to(dt)
                tmp = torch.empty(out.shape, dtype=dt, device=out.device) if (lf32 is not None and not fwd) else out
                _lib.check(_lib.lib().ffc_outer_pass_all(pr.handle, _DT[dt], int(fwd), _lib.ptr(inp), _lib.ptr(tmp), _lib.ptr(gate), bv,
                                                         npair, Hin, mi, Llong, ctypes.c_float(scale), _lib.stream_ptr()), "ffc_outer_pass_all (wide)")
                if tmp is not out:
                    out.copy_(tmp)
                return
            if _ONE_LAUNCH_LEVEL:       # all R passes in one launch (long side read / written once)
                _lib.check(_lib.lib().ffc_outer_pass_all(pr.handle, dcode, int(fwd), _lib.ptr(inp), _lib.ptr(out), _lib.ptr(gate), bv,
                                                         npair, Hin, mi, Llong, ctypes.c_float(scale), _lib.stream_ptr()), "ffc_outer_pass_all")
                return
            for c in range(R):
                _lib.check(_lib.lib().ffc_outer_pass_r(pr.handle, c, dcode, int(fwd), _lib.ptr(inp), _lib.ptr(out), _lib.ptr(gate), bv,
                                                       npair, Hin, mi, Llong, ctypes.c_float(scale), _lib.stream_ptr()), "ffc_outer_pass_r")
            return
        p16, p32 = self._plan(16384), self._plan(32768)
        _lib.check(_lib.lib().ffc_outer_pass(p16.handle, p32.handle, n0, dcode, int(fwd), _lib.ptr(inp), _lib.ptr(out),
                                             _lib.ptr(gate), bv, npair, Hin, mi, Llong, ctypes.c_float(scale),
                                             _lib.stream_ptr()), "ffc_outer_pass")

    def k_prescale(self, dt):
        return 256.0 if dt == torch.float16 else 1.0

    def to_dtype_rows(self, dt, k, H, Lk, pre=1.0):
        # one elementwise kernel (fp32 product rounded once into the dtype), not multiply + cast + copy
        k = k.detach().reshape(1, H, Lk)
        if pre == 1.0:
            return k.to(dt).contiguous()
        out = torch.empty(1, H, Lk, dtype=dt, device=k.device)
        return torch.mul(k, pre, out=out)

    def to_float_rows(self, out, H, Lk):
        return out[0].float()

    def kfft_c(self, dt, M, x, hp, scale):
        plan = self._plan(M)
        kf = torch.empty(hp, plan.kf_elems, 2, dtype=dt, device=self.device)
        _lib.check(_lib.lib().ffc_kernel_fft_c(plan.handle, _lib.ptr(x), hp, _lib.ptr(kf), ctypes.c_float(scale),
                                               _lib.stream_ptr()), "ffc_kernel_fft_c")
        return kf

    def conv(self, dt, M, x, kf, conj):
        return _conv(self._plan(M), x, kf, None, None, conj)

    def dkf(self, dt, M, xd, xu):
        plan = self._plan(M)
        Bp, hp, _ = xu.shape
        lib = _lib.lib()
        ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, Bp, hp), dtype=torch.uint8, device=self.device)
        _lib.check(lib.ffc_conv_bwd_dkf(plan.handle, _lib.ptr(xd), _lib.ptr(xu), None, None, _lib.ptr(ws), Bp, hp, M,
                                        _lib.stream_ptr()), "ffc_conv_bwd_dkf")
        return ws

    def conv_kx(self, dt, M, x, xk, scale, keep):
        """inner forward in ONE call (ffc_conv_fwd_kx): k_f rows from their complex input xk (inside the convolution launch where a
        workgroup owns its row), the convolution of x, spectra kept when `keep` -> (y, k_f, z or None)"""
        plan = self._plan(M)
        Bp, hp, _ = x.shape
        kf = torch.empty(hp, plan.kf_elems, 2, dtype=dt, device=self.device)
        # (the module-level budget test of the HBM-level path already charged x + z + y: no second charge here)
        z = _spectrum_buffer(plan, Bp, hp, self.device, True, "always") if keep else None
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ffc_conv_fwd_kx(plan.handle, _lib.ptr(xk), ctypes.c_float(scale), _lib.ptr(kf), _lib.ptr(x), _lib.ptr(y),
                                              _lib.ptr(z), Bp, hp, M, _lib.stream_ptr()), "ffc_conv_fwd_kx")
        return y, kf, z

    def bwd_dk(self, dt, M, xd, xu, kf, z, scale):
        """inner backward in ONE call (ffc_conv_bwd_kx): input-gradient rows + the dk rows as a complex pair-plane tensor (bf16)"""
        plan = self._plan(M)
        Bp, hp, _ = xd.shape
        lib = _lib.lib()
        ws = torch.empty(_ws_bytes(plan, Bp, hp), dtype=torch.uint8, device=self.device)
        yd = torch.empty_like(xd)
        out = torch.empty(2, hp, M, dtype=torch.bfloat16, device=self.device)
        _lib.check(lib.ffc_conv_bwd_kx(plan.handle, _lib.ptr(xd), _lib.ptr(xu), _lib.ptr(kf), _lib.ptr(yd), _lib.ptr(ws), _lib.ptr(z),
                                       _lib.ptr(out), ctypes.c_float(scale), Bp, hp, M, _lib.stream_ptr()), "ffc_conv_bwd_kx")
        return yd, out

    def conv_save(self, dt, M, x, kf):
        """inner forward that also keeps the inner spectra (None when the inner plan has no such path)"""
        plan = self._plan(M)
        z = _spectrum_buffer(plan, x.shape[0], x.shape[1], self.device, True, "always")      # charged once, at module level
        return (_conv(plan, x, kf, None, None, False), None) if z is None else (_conv_save(plan, x, kf, None, None, z), z)

    def bwd(self, dt, M, xd, xu, kf, z=None):
        """fused inner backward on pair-plane rows: (input gradient rows, fp32 dk_f slabs); z = spectra kept by conv_save"""
        plan = self._plan(M)
        Bp, hp, _ = xd.shape
        lib = _lib.lib()
        ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, Bp, hp), dtype=torch.uint8, device=self.device)
        yd = torch.empty_like(xd)
        if z is not None:
            _lib.check(lib.ffc_conv_bwd_z(plan.handle, _lib.ptr(xd), _lib.ptr(xu), _lib.ptr(kf), None, None, _lib.ptr(yd), None, None,
                                          _lib.ptr(ws), _lib.ptr(z), Bp, hp, M, 0, 0, 0, 0, 0, 0, 0, _lib.stream_ptr()), "ffc_conv_bwd_z")
        else:
            _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(xd), _lib.ptr(xu), _lib.ptr(kf), None, None, _lib.ptr(yd), None, _lib.ptr(ws),
                                        Bp, hp, M, _lib.stream_ptr()), "ffc_conv_bwd")
        return yd, ws

    def dkifft_c(self, M, ws, Bp, hp, scale, nslab=None):
        plan = self._plan(M)
        out = torch.empty(2, hp, M, dtype=torch.bfloat16, device=self.device)
        if nslab is None:
            _lib.check(_lib.lib().ffc_kernel_ifft_grad_c(plan.handle, _lib.ptr(ws), Bp, hp, _lib.ptr(out), ctypes.c_float(scale),
                                                         _lib.stream_ptr()), "ffc_kernel_ifft_grad_c")
        else:      # caller-owned slabs (B-shard: the reduce-scattered rows of this rank's heads)
            _lib.check(_lib.lib().ffc_kernel_ifft_grad_c_slabs(plan.handle, _lib.ptr(ws), nslab, hp, _lib.ptr(out), ctypes.c_float(scale),
                                                               _lib.stream_ptr()), "ffc_kernel_ifft_grad_c_slabs")
        return out


def _big_kernel_fft(mod, k, fac=None):
    """k (h, Lk) -> inner k_f rows (h * prod(N0), kf_elems, 2), head-major (heads can be sharded / gathered along dim 0)"""
    ops = _TorchOps(mod, k.device)
    return _big.kernel_fft(ops, mod.dtype, mod.seqlen, k.detach().to(torch.float32).contiguous(), k.shape[0], k.shape[-1], fac)


def _big_kf_mask(mod, fac, device, dtype):
    """frequency-sparse convolution at an HBM-level size: 0/1 mask over the inner k_f rows of one head, (rows, kf_elems), keeping
    the natural frequencies |f| < mod._kf_keep (bigfft.row_freq maps (row, inner position) to f)"""
    fac = fac or _big.BIG_FACTORS[mod.seqlen]
    key = ("big", mod.seqlen, fac, torch.device(device).index, dtype, mod._kf_keep)
    m = mod._masks.get(key)
    if m is None:
        import numpy as np
        plan = mod._get_plan(device, fac[1])
        idx = np.empty(plan.kf_elems, dtype=np.int32)
        _lib.check(_lib.lib().ffc_plan_kf_index(plan.handle, idx.ctypes.data_as(ctypes.c_void_p)), "ffc_plan_kf_index")
        offs, stride = _big.row_freq(mod.seqlen, fac)
        N = mod.seqlen
        f = (np.asarray(offs, np.int64)[:, None] + stride * idx.astype(np.int64)[None, :]) % N
        keep = (idx[None, :] >= 0) & ((f < mod._kf_keep) | (f > N - mod._kf_keep))
        m = torch.from_numpy(keep.astype(np.float32)).to(device=device, dtype=dtype)
        mod._masks[key] = m
    return m


# FFC_BIG_HALF=0: a batch of one runs the pair form with an empty partner row, as every odd last row still does (A/B runs)
_BIG_HALF = _os.environ.get("FFC_BIG_HALF", "1") != "0"


def _big_half(mod, B, Lmax, fac):
    """a batch of ONE row per head through a single HBM level: the half-row form (bigfft.rows_of) -- not for the frequency-sparse modules,
    whose masks are laid out over all inner rows"""
    factors, M = fac or _big.BIG_FACTORS[mod.seqlen]
    if len(factors) != 1 or _big.is_wide(factors[0], mod.seqlen // factors[0], Lmax):      # (the level's wide form stores all rows)
        return False
    return _BIG_HALF and B == 1 and mod._kf_keep is None


def _big_forward(mod, u, k, pregate, postgate, keep=False, kf=None, fac=None, half=False):
    """keep (training, module.save_spectrum): also return what the backward pass would otherwise compute again -- the
    transformed input x of the inner size (pair-plane rows), the inner spectra z (inner plans with that path) and, for the
    gated form, the inner output y (dpostgate = its inverse levels * dout).  kf: inner k_f rows computed elsewhere
    (_big_kernel_fft; the B-shard gathers them from the ranks), k is then unused."""
    N, dt = mod.seqlen, mod.dtype
    ops = _TorchOps(mod, u.device, half)
    B, H, L = u.shape
    M = (fac or _big.BIG_FACTORS[N])[1]
    z = None
    if kf is None and mod._kf_keep is None and _BIG_ONE_CALL:
        # one call for the inner k_f rows + the inner convolution (round 4: the last k -> k_f transform runs inside the convolution
        # launch where a workgroup owns its row, ffc_conv_fwd_kx)
        xk, sc = _big.kernel_rows(ops, dt, N, k.detach().to(torch.float32).contiguous(), k.shape[0], k.shape[-1], fac)
        x = _big.levels_forward(ops, dt, N, u, B, H, L, pregate, fac)
        y, kf, z = ops.conv_kx(dt, M, x, xk, sc, keep)
    else:
        if kf is None:
            kf = _big.kernel_fft(ops, mod.dtype, N, k.detach().to(torch.float32).contiguous(), k.shape[0], k.shape[-1], fac)
            if mod._kf_keep is not None:       # frequency-sparse k_f (flashfftconv/sparse_conv.py): zero the inner rows' bins |f| >= keep
                m = _big_kf_mask(mod, fac, u.device, kf.dtype)
                kf.view(H, m.shape[0], m.shape[1], 2).mul_(m[None, :, :, None])
        x = _big.levels_forward(ops, dt, N, u, B, H, L, pregate, fac)
        if keep:
            y, z = ops.conv_save(dt, M, x, kf)
        else:
            y = ops.conv(dt, M, x, kf, False)
    out = torch.empty_like(u)
    _big.levels_inverse(ops, dt, N, y, out, B, H, L, postgate, None, fac)
    # kept for the backward pass: the inner spectra z; the inner input rows x only when there are no spectra (the saved-spectra inner
    # kernel never reads them: round 5, a third of the kept bytes less); gated: the inner output y
    return out, kf, ((x if z is None else None, z, y if pregate is not None else None) if keep else None)


def _big_dk_from_dkf(mod, dkf, k_len, fac=None):
    """summed inner dk_f rows (h * prod(N0), kf_elems, 2) fp32 (one slab) -> dk (h, k_len) fp32"""
    ops = _TorchOps(mod, dkf.device)
    hp = dkf.shape[0]
    h = hp
    for n0 in (fac or _big.BIG_FACTORS[mod.seqlen])[0]:
        h //= n0
    return _big.dk_from_slabs(ops, mod.seqlen, dkf.contiguous(), 2, h, k_len, nslab=1, fac=fac)


def _big_backward(mod, dout, u, kf, pregate, postgate, k_len, kept=None, want_dkf=False, fac=None, half=False):
    """want_dkf: return the fp32 inner dk_f rows summed over the local batch (hp, kf_elems, 2) instead of dk (B-shard: the
    ranks reduce-scatter them and invert their own heads, _big_dk_from_dkf)"""
    N, dt = mod.seqlen, mod.dtype
    ops = _TorchOps(mod, u.device, half)
    B, H, L = u.shape
    M = (fac or _big.BIG_FACTORS[N])[1]
    xd = _big.levels_forward(ops, dt, N, dout, B, H, L, postgate, fac)
    xu, z, yu = kept if kept is not None else (_big.levels_forward(ops, dt, N, u, B, H, L, pregate, fac), None, None)
    if xu is None:
        if z is None or (pregate is not None and yu is None):      # (cannot happen with what _big_forward keeps)
            xu = _big.levels_forward(ops, dt, N, u, B, H, L, pregate, fac)
        # else: xu stays None -- with the inner spectra the (ungated) inner kernel does not read its input rows, and the C side accepts a
        # null `u` exactly then (conv_bwd_impl; ADVICE r05: a stand-in pointer would have hidden a kernel that does read it)
    # one fused inner launch (input gradient rows + fp32 dk_f partial sums; two transforms per pair on kept spectra, three
    # otherwise) instead of the dk_f kernel and the conj(k_f) forward kernel side by side (four)
    if not want_dkf and mod._kf_keep is None and _BIG_ONE_CALL:
        # one call: the fused inner backward + the inner dk_f -> complex rows step (out of the same launch where a workgroup owns its row)
        yd, ypair = ops.bwd_dk(dt, M, xd, xu, kf, z, _big.dk_pair_scale(N, fac))      # (xu None: spectra kept)
        dk = _big.dk_from_pair(ops, N, ypair, H, k_len, fac)
        return _big_backward_tail(mod, ops, dt, N, yd, dk, u, dout, xu, yu, kf, pregate, B, H, L, M, fac)
    yd, ws = ops.bwd(dt, M, xd, xu, kf, z)
    if xu is None:
        xu = xd          # (shape only, below)
    if mod._kf_keep is not None:
        # d/dk of (mask * FFT(k)): mask the fp32 inner dk_f partial sums (same row / position order as k_f) before the inverse
        plan_m = ops._plan(M)
        hp_m = xu.shape[1]
        nsl = _lib.lib().ffc_dkf_slab_count(plan_m.handle, xu.shape[0], hp_m)
        m32 = _big_kf_mask(mod, fac, u.device, torch.float32)
        ws[: nsl * hp_m * plan_m.kf_elems * 8].view(torch.float32).view(nsl, H, m32.shape[0], m32.shape[1], 2).mul_(m32[None, None, :, :, None])
    if want_dkf:
        plan = ops._plan(M)
        hp, nfl = xu.shape[1], xu.shape[1] * plan.kf_elems * 2
        nslab = _lib.lib().ffc_dkf_slab_count(plan.handle, xu.shape[0], hp)
        slabs = ws[: nslab * nfl * 4].view(torch.float32).view(nslab, hp, plan.kf_elems, 2)
        dk = slabs[0] if nslab == 1 else slabs.sum(0)
    else:
        dk = _big.dk_from_slabs(ops, N, ws, xu.shape[0], H, k_len, None, fac)
    return _big_backward_tail(mod, ops, dt, N, yd, dk, u, dout, xu, yu, kf, pregate, B, H, L, M, fac)


def _big_backward_tail(mod, ops, dt, N, yd, dk, u, dout, xu, yu, kf, pregate, B, H, L, M, fac):
    """the inverse levels of the gradients: du (* pregate), and for the gated form dpregate, dpostgate"""
    du = torch.empty_like(u)
    shared = {}
    _big.levels_inverse(ops, dt, N, yd, du, B, H, L, pregate, shared, fac)
    if pregate is None:
        return du, dk, None, None
    dpre = torch.empty_like(u)
    _big.levels_inverse(ops, dt, N, yd, dpre, B, H, L, u, shared, fac)
    if yu is None:
        yu = ops.conv(dt, M, xu, kf, False)
    dpost = torch.empty_like(u)
    _big.levels_inverse(ops, dt, N, yu, dpost, B, H, L, dout, None, fac)
    return du, dk, dpre, dpost


def _check_inputs(mod, u, k, gates):
    if not u.is_cuda:
        raise RuntimeError("FlashFFTConv: u must be a CUDA/HIP tensor (no CPU fallback in the product path)")
    if k.device != u.device:
        raise RuntimeError(f"FlashFFTConv: k is on {k.device}, u on {u.device}")
    for g in gates:
        if g is not None and g.device != u.device:
            raise RuntimeError(f"FlashFFTConv: gate is on {g.device}, u on {u.device}")
    if u.dim() != 3:
        raise RuntimeError("FlashFFTConv: u must be (B, H, L)")
    if u.dtype != mod.dtype:
        raise RuntimeError(f"FlashFFTConv: u.dtype {u.dtype} != module dtype {mod.dtype}")
    B, H, L = u.shape
    if L > mod.seqlen:
        raise RuntimeError(f"FlashFFTConv: L={L} exceeds fft size {mod.seqlen}")
    if k.dim() != 2 or k.shape[0] != H or k.shape[-1] > mod.seqlen:
        raise RuntimeError("FlashFFTConv: k must be (H, Lk) with Lk <= fft size")
    for g in gates:
        if g is not None and (g.shape != u.shape or g.dtype != u.dtype):
            raise RuntimeError("FlashFFTConv: gates must match u in shape and dtype")


class _FlashFFTConvFn(torch.autograd.Function):
    # reference: FlashFFTConvFunc (conv.py:563) and GatedFlashFFTConvFunc (conv.py:3236)

    @staticmethod
    def forward(ctx, u, k, mod, pregate, postgate):
        _check_inputs(mod, u, k, (pregate, postgate))
        with _dev_ctx(u.device):      # launches go to u's device and its current stream
            return _FlashFFTConvFn._forward(ctx, u, k, mod, pregate, postgate)

    @staticmethod
    def _forward(ctx, u, k, mod, pregate, postgate):
        u = u.contiguous()
        pregate = None if pregate is None else pregate.contiguous()
        postgate = None if postgate is None else postgate.contiguous()
        ctx.mod, ctx.k_len, ctx.k_dtype, ctx.gated = mod, k.shape[-1], k.dtype, pregate is not None
        ctx.big = mod._big or mod._route_big(max(u.shape[-1], k.shape[-1]))
        kept = None
        if ctx.big:
            keep = mod.training and mod.save_spectrum and _recording() and any(ctx.needs_input_grad[i] for i in (0, 1, 3, 4))
            if keep:      # kept: the inner spectra z (4 B per point and pair), gated also the inner output y (4 B)
                keep = _spectrum_budget_ok(((u.shape[0] + 1) // 2) * u.shape[1] * mod.seqlen * (8 if ctx.gated else 4), u.device, mod.save_spectrum)
                if not keep:
                    SPECTRUM_FALLBACKS["budget"] += 1
            # the factorisation may depend on the lengths (fft 4M: one level of 128 when everything fits a quarter of it)
            ctx.fac = fac = _big.choose(mod.seqlen, max(u.shape[-1], k.shape[-1]), _TorchOps)
            ctx.half = half = _big_half(mod, u.shape[0], max(u.shape[-1], k.shape[-1]), fac)
            try:
                out, kf, kept = _big_forward(mod, u, k, pregate, postgate, keep, None, fac, half)
            except torch.cuda.OutOfMemoryError:
                if not keep:
                    raise
                SPECTRUM_FALLBACKS["oom"] += 1
                out, kf, kept = _big_forward(mod, u, k, pregate, postgate, False, None, fac, half)
        else:
            plan = mod._get_plan(u.device, mod._plan_seqlen)
            kf = mod._cached_kf(k) if mod.cache_kf and not k.requires_grad else None
            # one C-ABI call for k -> k_f + the convolution (ffc_conv_fwd_k) unless something sits between the two
            one_call = kf is None and not mod._folded and mod._kf_keep is None
            if kf is None and not one_call:
                kf = _kernel_fft(plan, _periodise_k(k, mod.seqlen) if mod._folded else k)
                if mod._kf_keep is not None:       # frequency-sparse k_f (flashfftconv/sparse_conv.py)
                    kf.mul_(mod._kf_mask(plan, kf.dtype)[None, :, None])
                if mod.cache_kf and not k.requires_grad:
                    mod._kf_cache = (_kf_key(k), kf)
            # MI355X design (memory laid out for 288 GB of HBM): a training forward keeps every pair's spectrum
            # FFT(u * pregate) (2x the bytes of u at L = N/2) so that the backward pass does not transform u a second time: its
            # kernel runs two transforms per pair instead of three (fused backward -30 % at B16 H768 fft 32768,
            # profiles/r03_spectrum.txt).  The gated form also keeps the output before the postgate multiply (1x the bytes
            # of u): dpostgate = dout * that, instead of one more inverse transform of the spectrum.
            # module.save_spectrum = False (or FFC_SAVE_SPECTRUM=0) keeps the reference's recomputing backward.
            z = yraw = None
            rows = 0 if one_call else _sparse_rows(mod, plan)      # low-pass k_f: the forward kernel that skips the all-zero spectrum rows
            if rows:
                out = _conv_sparse(plan, u, kf, pregate, postgate, False, rows)
            elif mod.training and mod.save_spectrum and _recording() and any(ctx.needs_input_grad[i] for i in (0, 1, 3, 4)):
                # Gated single-tile sizes, opt-in (fft <= _Y_ONLY_MAX, round 6): keep ONLY the output before the postgate; the backward
                # transforms u * pregate again (see _Y_ONLY_MAX: a third less kept memory at the same fwd + bwd time +- 3 %).
                y_only = ctx.gated and plan.seqlen <= _Y_ONLY_MAX
                if not y_only:
                    z = _spectrum_buffer(plan, u.shape[0], u.shape[1], u.device, ctx.gated, mod.save_spectrum)
                if (z is not None or y_only) and ctx.gated:
                    try:
                        yraw = torch.empty_like(u)
                    except torch.cuda.OutOfMemoryError:
                        z = None
            if one_call:
                B, H, L = u.shape
                k32 = k if (k.dtype == torch.float32 and k.is_contiguous()) else k.detach().to(torch.float32).contiguous()
                kf = torch.empty(H, plan.kf_elems, 2, dtype=plan.dtype, device=u.device)
                out = torch.empty_like(u)
                _lib.check(_lib.lib().ffc_conv_fwd_k(plan.handle, _lib.ptr(k32), k32.shape[-1], _lib.ptr(kf), _lib.ptr(u), _lib.ptr(pregate),
                                                     _lib.ptr(postgate), _lib.ptr(out), _lib.ptr(z), _lib.ptr(yraw), B, H, L,
                                                     _lib.stream_ptr()), "ffc_conv_fwd_k")
                if mod.cache_kf and not k.requires_grad:
                    mod._kf_cache = (_kf_key(k), kf)
            elif not rows:
                out = (_conv(plan, u, kf, pregate, postgate, False) if z is None and yraw is None
                       else _conv_save(plan, u, kf, pregate, postgate, z, yraw))
        if mod.training:  # reference saves for backward only in training mode (conv.py:587-588)
            # (z, yraw: saved tensors, released with the graph and kept by retain_graph like the others)
            if ctx.big:
                extra = () if kept is None else tuple(t for t in kept if t is not None)
                ctx.kept_layout = None if kept is None else tuple(t is not None for t in kept)
            else:
                extra = tuple(t for t in (z, yraw) if t is not None)      # (z, yraw) | (z,) | (yraw,): gated single-tile sizes keep y_raw alone
                ctx.kept_z = z is not None
            ctx.save_for_backward(*(((u, kf, pregate, postgate) if ctx.gated else (u, kf)) + extra))
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.saved_tensors:
            raise RuntimeError("FlashFFTConv: backward needs module.training=True at forward time")
        with _dev_ctx(ctx.saved_tensors[0].device):
            return _FlashFFTConvFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        dout = dout.contiguous()
        z = yraw = None
        if ctx.gated:
            u, kf, pregate, postgate = ctx.saved_tensors[:4]
            if len(ctx.saved_tensors) > 4:
                z, yraw = ctx.saved_tensors[4:6] if getattr(ctx, "kept_z", True) else (None, ctx.saved_tensors[4])
        else:
            (u, kf), pregate, postgate = ctx.saved_tensors[:2], None, None
            z = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        if ctx.big:
            kept = None
            if getattr(ctx, "kept_layout", None) is not None:
                it = iter(ctx.saved_tensors[4 if ctx.gated else 2:])
                kept = tuple(next(it) if present else None for present in ctx.kept_layout)
            du, dk, dpre, dpost = _big_backward(ctx.mod, dout, u, kf, pregate, postgate, ctx.k_len, kept, False, ctx.fac, ctx.half)
            return du, dk.to(ctx.k_dtype), None, dpre, dpost
        plan = ctx.mod._get_plan(u.device, ctx.mod._plan_seqlen)
        B, H, L = u.shape
        lib = _lib.lib()
        k_len = plan.seqlen if ctx.mod._folded else ctx.k_len
        # one fused launch: du (+ dpregate, dpostgate when gated) + fp32 dk_f partial sums; then dk_f -> dk
        ws = torch.empty(_ws_bytes(plan, B, H), dtype=torch.uint8, device=u.device)
        du = torch.empty_like(u)
        dpre = torch.empty_like(u) if ctx.gated else None
        dpost = torch.empty_like(u) if ctx.gated else None
        if not ctx.mod._folded and ctx.mod._kf_keep is None:
            # one C-ABI call: fused backward (on the saved spectra when there are any) + dk_f -> dk (ffc_conv_bwd_k)
            dk = torch.empty(H, k_len, dtype=torch.float32, device=u.device)
            _lib.check(lib.ffc_conv_bwd_k(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pregate), _lib.ptr(postgate),
                                          _lib.ptr(du), _lib.ptr(dpre), _lib.ptr(dpost), _lib.ptr(ws), _lib.ptr(z), _lib.ptr(yraw), _lib.ptr(dk),
                                          k_len, B, H, L, _lib.stream_ptr()), "ffc_conv_bwd_k")
            if dk.dtype != ctx.k_dtype:
                dk = dk.to(ctx.k_dtype)
            return (du, dk, None, dpre, dpost) if ctx.gated else (du, dk, None, None, None)
        if (z is not None or yraw is not None) and ctx.gated:
            # dpostgate = dout * y_raw out of the kernel's dout row load (round 3: a torch elementwise kernel, 3 x |u| bytes more)
            _lib.check(lib.ffc_conv_bwd_zy(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pregate),
                                           _lib.ptr(postgate), _lib.ptr(du), _lib.ptr(dpre), _lib.ptr(dpost), _lib.ptr(ws), _lib.ptr(z),
                                           _lib.ptr(yraw), B, H, L, 0, 0, 0, 0, 0, 0, 0, _lib.stream_ptr()), "ffc_conv_bwd_zy")
        elif z is not None:
            _lib.check(lib.ffc_conv_bwd_z(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pregate),
                                          _lib.ptr(postgate), _lib.ptr(du), _lib.ptr(dpre), None, _lib.ptr(ws), _lib.ptr(z),
                                          B, H, L, 0, 0, 0, 0, 0, 0, 0, _lib.stream_ptr()), "ffc_conv_bwd_z")
        else:
            _lib.check(lib.ffc_conv_bwd_gated(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), _lib.ptr(pregate),
                                              _lib.ptr(postgate), _lib.ptr(du), _lib.ptr(dpre), _lib.ptr(dpost), _lib.ptr(ws),
                                              B, H, L, _lib.stream_ptr()), "ffc_conv_bwd_gated")
        if ctx.mod._kf_keep is not None:
            # d/dk of (mask * FFT(k)): mask the fp32 dk_f partial sums (same internal order as k_f) before the inverse
            nfl = H * plan.kf_elems * 2
            slabs = ws[: (ws.numel() // 4 // nfl) * nfl * 4].view(torch.float32).view(-1, H, plan.kf_elems, 2)
            nslab = lib.ffc_dkf_slab_count(plan.handle, B, H)
            slabs[:nslab].mul_(ctx.mod._kf_mask(plan, torch.float32)[None, None, :, None])
        dk = torch.empty(H, k_len, dtype=torch.float32, device=u.device)
        _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, k_len, _lib.ptr(dk), _lib.stream_ptr()),
                   "ffc_kernel_ifft_grad")
        if ctx.mod._folded:      # chain rule of _periodise_k
            n = ctx.mod.seqlen
            dk = (dk[:, :n] + dk[:, n:])[:, :ctx.k_len]
        dk = dk.to(ctx.k_dtype)
        if not ctx.gated:
            return du, dk, None, None, None
        return du, dk, None, dpre, dpost


class FlashFFTConv(torch.nn.Module):
    """reference: flashfftconv/conv.py:71-560.

    Differences a caller can observe:
      * `use_32_butterfly` is accepted for signature compatibility and has no effect: in the reference it only selects
        which of two equivalent factorisations (16- or 32-point outer butterfly) an fft size >= 65536 uses
        (conv.py:262-551); here the factorisation is internal to the plan (multi-pass kernels up to 131072,
        flashfftconv/bigfft.py above) and the result is the same convolution either way.
      * no tables are registered as buffers: `state_dict()` of this module is empty.  Checkpoints saved from the
        reference module carry its DFT / twiddle buffers (`*.f_32_fft`, `*.twiddle_factors_fft_32_1K`, ...); they are
        accepted and discarded on load, so `load_state_dict(strict=True)` of a reference checkpoint works.
      * one tensor must stay below 2^31 elements (B*H*L, and for fft sizes >= 262144 also 2*ceil(B/2)*H*fft_size, the
        complex intermediate): larger calls raise RuntimeError; split the batch.
      * H % 16 == 0 is NOT required for fft sizes > 32768 (reference README.md:269), L may be any length <= fft size.
      * seqlen > 32768 and rows whose linear convolution fits a smaller fft size (Lu + Lk - 1 <= seqlen / 2): that size runs
        (_fit_seqlen) -- the same outputs and gradients, nothing wraps at either size; `fit_fft = False` runs seqlen points."""

    def __init__(self, seqlen, dtype=torch.float16, use_32_butterfly=True):
        super().__init__()
        assert dtype == torch.bfloat16 or dtype == torch.float16
        if seqlen not in SUPPORTED_SEQLENS:
            raise NotImplementedError(f"seqlen {seqlen} not supported")
        self.seqlen = seqlen
        self._big = seqlen in _big.BIG_FACTORS and seqlen not in MULTIPASS_SEQLENS
        # fft 2048: 2 passes of the 1024 kernel (multi-pass plan of its own); folded onto the 4096 plan only when 2048 is
        # taken out of FFC_MULTIPASS (A/B runs)
        self._folded = seqlen in FOLDED_SEQLENS and seqlen not in MULTIPASS_SEQLENS
        self._plan_seqlen = FOLDED_SEQLENS.get(seqlen, seqlen) if self._folded else seqlen
        self.dtype = dtype
        self.use_32_butterfly = use_32_butterfly      # no effect, see the class docstring
        self._kf_keep = None        # frequency-sparse mode: keep bins |f| < _kf_keep (set by sparse_conv)
        self._masks = {}
        # Opt-in inference cache of k_f (the reference recomputes FFT(k) in every forward,
        # conv.py:572-575; its users hand-roll kernel caching, examples/bert/README.md).  When True and `k` does not
        # require grad, k_f is reused while k is the same storage at the same version.  Off by default: a filter
        # that is re-generated into recycled memory every step would look "unchanged".
        self.cache_kf = False
        self._kf_cache = None
        # training forward keeps FFT(u) for the backward pass (see _FlashFFTConvFn._forward); FFC_SAVE_SPECTRUM=0 turns it off
        # True: when the buffer fits the memory budget (_spectrum_budget_ok: at most 1/8 of the free memory at that moment);
        # "always" (FFC_SAVE_SPECTRUM=always): whenever the allocation succeeds; False: never (the reference's footprint)
        _sv = _os.environ.get("FFC_SAVE_SPECTRUM", "1")
        self.save_spectrum = False if _sv == "0" else ("always" if _sv == "always" else True)
        # rows much shorter than the fft size run on the smallest fft size that holds their linear convolution (_fit_seqlen);
        # False (or FFC_FIT_FFT=0) always runs `seqlen` points
        self.fit_fft = True
        self._fitted = {}

    def _fit_seqlen(self, Lu, Lk):
        """The fft size a call with rows of Lu (u, gates, output) and Lk (k) samples runs on.  While Lu + Lk - 1 <= n the n-point
        circular convolution does not wrap: its first Lu outputs, and du / dk of them, are the LINEAR convolution's -- the same numbers
        for every such n, `seqlen` included (one rounding stage fewer per halving).  A module built for the longest sequence of a
        model (the reference's callers fix fft_size = 2 * l_max at construction, examples/hyena-dna) and called with shorter rows, or
        BASELINE config 4 (fft 4194304 around L = 1048576: 2097152 points hold it), therefore runs the smallest supported size
        n >= Lu + Lk - 1.  Only for seqlen > 32768, where a halving saves a kernel pass or an HBM level (below that the fused kernel's
        implicit zero padding already skips the row traffic), and not for the frequency-sparse modules (their masks are defined on the
        seqlen-point spectrum)."""
        n = self.seqlen
        if not (_FIT_FFT and getattr(self, "fit_fft", True)) or self._kf_keep is not None or n <= 32768:
            return n
        need = max(Lu + Lk - 1, 1)
        while n > 256 and n // 2 >= need:
            n //= 2
        return n

    def _fitted_module(self, n):
        #  created lazily through getattr: a module unpickled / deep-copied from a version without the attribute still
        # forwards; the children follow the parent's mode at every call and drop their cached k_f when the parent has none (cache_kf
        # off, or invalidated), so that at most the sizes in use keep a k_f alive.  Not thread-safe: two threads that call ONE module in
        # different modes race on the child's flags exactly as they would on the parent's own `training` attribute.
        fitted = self.__dict__.setdefault("_fitted", {})
        m = fitted.get(n)
        if m is None:
            m = fitted[n] = FlashFFTConv(n, dtype=self.dtype, use_32_butterfly=self.use_32_butterfly)
        m.training, m.save_spectrum, m.cache_kf, m.fit_fft = self.training, self.save_spectrum, self.cache_kf, False
        if not self.cache_kf or self._kf_cache is None:
            for c in fitted.values():
                if c is not m:
                    c._kf_cache = None
        return m

    def graphed_step(self, u, k, dout, pregate=None, postgate=None, warmup=3):
        """forward + backward of this module captured into ONE HIP graph on static copies of the given tensors
        (flashfftconv/graphs.py GraphedStep): for the short sequences, where a step is bound by the host side of autograd and
        of the launches, not by its 30 - 50 us of kernels.  step(u, k, dout) -> (y, du, dk[, dpregate, dpostgate])."""
        from .graphs import GraphedStep
        return GraphedStep(self, u, k, dout, pregate, postgate, warmup)

    def _route_big(self, Lmax):
        """Per-call routing of fft 131072 (round 4, profiles/r04_route.txt, B16 H768 / H384, fwd + bwd ms): 4 passes of the fused 32768
        kernel win while the rows fit half the fft size (L = 32K: 8.9 vs 13.3, L = 64K: 13.65 vs 13.5 at a third of the memory), one
        HBM level around the fused 4096 kernel wins for longer rows (L = 128K: 7.1 vs 12.6 -- every pass of the multi-pass form
        re-reads all four input blocks).  fft 65536 stays on its 2 passes at every length (L = 64K: 6.7 vs 7.2)."""
        return self.seqlen == 131072 and not self._big and Lmax > 65536 and 131072 in MULTIPASS_SEQLENS and _ROUTE_BY_LENGTH

    def _cached_kf(self, k):
        c = self._kf_cache
        return c[1] if c is not None and c[0] == _kf_key(k) else None

    def _kf_mask(self, plan, dtype):
        """0/1 mask over k_f's internal positions keeping the natural frequencies |f| < self._kf_keep."""
        key = (plan.seqlen, plan.device.index, dtype, self._kf_keep)
        m = self._masks.get(key)
        if m is None:
            import numpy as np
            idx = np.empty(plan.kf_elems, dtype=np.int32)
            _lib.check(_lib.lib().ffc_plan_kf_index(plan.handle, idx.ctypes.data_as(ctypes.c_void_p)), "ffc_plan_kf_index")
            f = idx.astype(np.int64)
            N = plan.seqlen
            keep = (f >= 0) & ((f < self._kf_keep) | (f > N - self._kf_keep))
            m = torch.from_numpy(keep.astype(np.float32)).to(device=plan.device, dtype=dtype)
            self._masks[key] = m
        return m

    def _get_plan(self, device, N=None):
        return get_plan(self.seqlen if N is None else N, self.dtype, device)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # swallow the reference module's table buffers (see _REF_BUFFER): a reference checkpoint loads strictly
        for key in [k for k in state_dict if k.startswith(prefix) and "." not in k[len(prefix):]
                    and _REF_BUFFER.match(k[len(prefix):])]:
            del state_dict[key]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, u, k, pregate=None, postgate=None):
        if pregate is not None or postgate is not None:
            assert pregate is not None and postgate is not None
        if self.seqlen > 32768 and torch.is_tensor(u) and torch.is_tensor(k) and u.dim() == 3 and k.dim() == 2:
            n = self._fit_seqlen(u.shape[-1], k.shape[-1])
            if n != self.seqlen:
                return self._fitted_module(n)(u, k, pregate, postgate)
        return _apply_noting_grad_mode(_FlashFFTConvFn, u, k, self, pregate, postgate)
