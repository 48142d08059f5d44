"""Fused Hyena / Monarch-Mixer sequence operator on the MI355X kernels.

The reference's callers build the gated long convolution out of separate kernels around FlashFFTConv
(examples/hyena-dna/hyenadna_flashfftconv.py:269-289, examples/bert/monarch_mixer_sequence_mixer_flashfftconv.py:118-175):

    uc = flash_short_filter(u)[..., :l]            # depthwise k=3 conv over the 3*D projected channels
    x1, x2, v = uc.split(d_model, dim=1)           # three (B, D, L) channel slices (non-contiguous)
    x1v = (x1 * v).contiguous()                    # elementwise kernel + copy
    y = flashfftconv(x1v, k)                       # FFT convolution
    y = y * x2                                     # elementwise kernel

Here the three slices are handed to the gated FFT-conv kernel IN PLACE (batch-strided rows, ffc_conv_fwd_strided): v is the
input, x1 the pregate, x2 the postgate, so the multiply kernels, the copy and their HBM round trips disappear, in the forward
and in the backward (du / dpregate / dpostgate are written straight into the slices of d(uc), one launch; then the short
convolution's own backward).  y = x2 * conv(x1 * v, k): the same function as the reference composition."""
import torch

from . import _lib
from .conv import FlashFFTConv, _check_inputs, _kernel_fft, _periodise_k, _spectrum_buffer, _kf_key, _apply_noting_grad_mode, _recording
from .depthwise_1d import FlashDepthWiseConv1d


def _slice_ptr(t, j, D, L):
    return _lib.ctypes.c_void_p(t.data_ptr() + j * D * L * t.element_size())


class _GatedSlicesFn(torch.autograd.Function):
    """y = uc[:, 1] * conv(uc[:, 0] * uc[:, 2], k) for uc viewed as (B, 3, D, L): x1 = slice 0, x2 = slice 1, v = slice 2."""

    @staticmethod
    def forward(ctx, uc, k, mod):
        B, D3, L = uc.shape
        D = D3 // 3
        plan = mod._get_plan(uc.device, mod._plan_seqlen)
        with torch.cuda.device(uc.device):
            kf = mod._cached_kf(k) if mod.cache_kf and not k.requires_grad else None      # inference cache, as FlashFFTConv
            if kf is None:
                kf = _kernel_fft(plan, _periodise_k(k, mod.seqlen) if mod._folded else k)
                if mod.cache_kf and not k.requires_grad:
                    mod._kf_cache = (_kf_key(k), kf)
            y = torch.empty(B, D, L, dtype=uc.dtype, device=uc.device)
            sb = D3 * L
            # training: keep the spectra FFT(x1 * v) and the output before the x2 multiply (FlashFFTConv.save_spectrum)
            z = yraw = None
            if mod.training and mod.save_spectrum and _recording() and any(ctx.needs_input_grad[:2]):
                z = _spectrum_buffer(plan, B, D, uc.device, True, mod.save_spectrum)
                if z is not None:
                    try:
                        yraw = torch.empty_like(y)
                    except torch.cuda.OutOfMemoryError:
                        z = None
            if z is None:
                _lib.check(_lib.lib().ffc_conv_fwd_strided(plan.handle, _slice_ptr(uc, 2, D, L), _lib.ptr(kf), _slice_ptr(uc, 0, D, L),
                                                           _slice_ptr(uc, 1, D, L), _lib.ptr(y), B, D, L, 0, sb, sb, sb, 0,
                                                           _lib.stream_ptr()), "ffc_conv_fwd_strided")
            else:
                _lib.check(_lib.lib().ffc_conv_fwd_z(plan.handle, _slice_ptr(uc, 2, D, L), _lib.ptr(kf), _slice_ptr(uc, 0, D, L),
                                                     _slice_ptr(uc, 1, D, L), _lib.ptr(y), _lib.ptr(z), _lib.ptr(yraw), B, D, L,
                                                     sb, sb, sb, 0, _lib.stream_ptr()), "ffc_conv_fwd_z")
        ctx.mod, ctx.k_len, ctx.k_dtype = mod, k.shape[-1], k.dtype
        if mod.training:
            ctx.save_for_backward(*((uc, kf) + (() if z is None else (z, yraw))))
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.saved_tensors:
            raise RuntimeError("FlashHyenaOp: backward needs module.training=True at forward time")
        uc, kf = ctx.saved_tensors[:2]
        z, yraw = ctx.saved_tensors[2:4] if len(ctx.saved_tensors) > 2 else (None, None)
        mod = ctx.mod
        B, D3, L = uc.shape
        D = D3 // 3
        plan = mod._get_plan(uc.device, mod._plan_seqlen)
        lib = _lib.lib()
        with torch.cuda.device(uc.device):
            dy = dy.contiguous()
            duc = torch.empty_like(uc)
            ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, D), dtype=torch.uint8, device=uc.device)
            sb = D3 * L
            # u = v (slice 2), pregate = x1 (slice 0), postgate = x2 (slice 1); gradients land in the same slices of duc
            if z is not None:
                # d x2 = dy * (output before the x2 multiply): written into its slice of duc by the kernel's dy row load
                _lib.check(lib.ffc_conv_bwd_zy(plan.handle, _lib.ptr(dy), _slice_ptr(uc, 2, D, L), _lib.ptr(kf),
                                               _slice_ptr(uc, 0, D, L), _slice_ptr(uc, 1, D, L), _slice_ptr(duc, 2, D, L),
                                               _slice_ptr(duc, 0, D, L), _slice_ptr(duc, 1, D, L), _lib.ptr(ws), _lib.ptr(z), _lib.ptr(yraw),
                                               B, D, L, 0, sb, sb, sb, sb, sb, sb, _lib.stream_ptr()), "ffc_conv_bwd_zy")
            else:
              _lib.check(lib.ffc_conv_bwd_gated_strided(plan.handle, _lib.ptr(dy), _slice_ptr(uc, 2, D, L), _lib.ptr(kf),
                                                      _slice_ptr(uc, 0, D, L), _slice_ptr(uc, 1, D, L), _slice_ptr(duc, 2, D, L),
                                                      _slice_ptr(duc, 0, D, L), _slice_ptr(duc, 1, D, L), _lib.ptr(ws), B, D, L,
                                                      0, sb, sb, sb, sb, sb, sb, _lib.stream_ptr()), "ffc_conv_bwd_gated_strided")
            k_len = plan.seqlen if mod._folded else ctx.k_len
            dk = torch.empty(D, k_len, dtype=torch.float32, device=uc.device)
            _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, D, k_len, _lib.ptr(dk), _lib.stream_ptr()),
                       "ffc_kernel_ifft_grad")
            if mod._folded:
                n = mod.seqlen
                dk = (dk[:, :n] + dk[:, n:])[:, :ctx.k_len]
        return duc, dk.to(ctx.k_dtype), None


def gated_conv_from_slices(conv, uc, k):
    """y = x2 * conv(x1 * v, k) with (x1, x2, v) = uc.split(D, dim=1), uc (B, 3D, L) contiguous, through `conv`
    (a FlashFFTConv).  Sizes served by the fused kernels (fft <= 131072) read the slices in place; larger ones fall back to
    the module's gated call on contiguous copies (same result)."""
    if uc.dim() != 3 or uc.shape[1] % 3:
        raise RuntimeError("gated_conv_from_slices: uc must be (B, 3*D, L)")
    D, L = uc.shape[1] // 3, uc.shape[2]
    # a model built for its longest sequence and run on a shorter one: the smallest fft size that holds the rows (FlashFFTConv._fit_seqlen)
    n = conv._fit_seqlen(L, k.shape[-1]) if k.dim() == 2 else conv.seqlen
    if n != conv.seqlen:
        conv = conv._fitted_module(n)
    # (the strided launchers address one tensor with 31-bit element offsets: (B-1) * 3*D*L + D*L has to stay below 2^31,
    # where the composition on contiguous copies only needs B*D*L < 2^31)
    too_wide = (uc.shape[0] - 1) * 3 * D * L + D * L >= 2 ** 31
    if conv._big or conv._route_big(max(L, k.shape[-1])) or conv._kf_keep is not None or (D * L) % 8 or not uc.is_contiguous() or too_wide:
        x1, x2, v = (t.contiguous() for t in uc.split(D, dim=1))
        return conv(v, k, x1, x2)
    _check_inputs(conv, uc[:, :D], k, ())
    return _apply_noting_grad_mode(_GatedSlicesFn, uc, k, conv)


class FlashHyenaOp(torch.nn.Module):
    """short depthwise conv (k = 3, "same" length) over the 3*D projected channels, then y = x2 * fftconv(x1 * v, k).

    forward(x1x2v, k): x1x2v (B, 3*D, L) bf16/fp16 (the in-projection's output, channels first), k (D, Lk) fp32 -> (B, D, L).
    `short_filter_weight` (3D, 1, 3) or (3D, 3) and `short_filter_bias` (3D,) are the nn.Conv1d parameters the reference
    callers pass to FlashDepthWiseConv1d (hyenadna_flashfftconv.py:248-261: padding=1, i.e. the [..., :l] crop is a no-op)."""

    def __init__(self, d_model, fft_size, short_filter_weight, short_filter_bias, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.d_model = d_model
        self.short_filter = FlashDepthWiseConv1d(3 * d_model, 3, 1, short_filter_weight, short_filter_bias, is_bhl=True,
                                                 device=device, dtype=dtype)
        self.flashfftconv = FlashFFTConv(fft_size, dtype=dtype)

    def forward(self, x1x2v, k):
        uc = self.short_filter(x1x2v)
        return gated_conv_from_slices(self.flashfftconv, uc, k)


_LOOP_MAX_BATCH = 16


class _ProjectIn(torch.autograd.Function):
    """(C, D) x (B, L, D) -> (B, C, L): one 2-D GEMM per batch row on the transposed view, forward and backward"""

    @staticmethod
    def forward(ctx, weight, u):
        B, L, _ = u.shape
        w = weight if weight.dtype == u.dtype else weight.to(u.dtype)      # fp32 master weights under mixed precision
        out = torch.empty(B, w.shape[0], L, dtype=u.dtype, device=u.device)
        for b in range(B):
            torch.mm(w, u[b].t(), out=out[b])
        ctx.save_for_backward(w, u)
        ctx.w_dtype = weight.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        weight, u = ctx.saved_tensors
        B = u.shape[0]
        du = dw = None
        if ctx.needs_input_grad[1]:
            du = torch.empty_like(u)
            for b in range(B):
                torch.mm(g[b].t(), weight, out=du[b])              # (L, C) x (C, D), g[b].t() read as a transposed operand
        if ctx.needs_input_grad[0]:
            dw = torch.mm(g[0], u[0])                              # (C, L) x (L, D)
            for b in range(1, B):
                dw.addmm_(g[b], u[b])
            dw = dw.to(ctx.w_dtype)
        return dw, du


class _ProjectOut(torch.autograd.Function):
    """(B, D, L) -> (B, L, C) = y[b]^T W^T + bias: one 2-D GEMM per batch row reading y[b].t() as its transposed operand"""

    @staticmethod
    def forward(ctx, weight, bias, y):
        B, _, L = y.shape
        w = weight if weight.dtype == y.dtype else weight.to(y.dtype)      # fp32 master weights under mixed precision
        bb = bias if bias is None or bias.dtype == y.dtype else bias.to(y.dtype)
        out = torch.empty(B, L, w.shape[0], dtype=y.dtype, device=y.device)
        wt = w.t()
        for b in range(B):
            if bb is None:
                torch.mm(y[b].t(), wt, out=out[b])
            else:
                torch.addmm(bb, y[b].t(), wt, out=out[b])
        ctx.save_for_backward(w, y)
        ctx.has_bias = bias is not None
        ctx.w_dtype, ctx.b_dtype = weight.dtype, (None if bias is None else bias.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        weight, y = ctx.saved_tensors
        B = y.shape[0]
        dw = db = dy = None
        if ctx.needs_input_grad[2]:
            dy = torch.empty_like(y)
            wt = weight.t()
            for b in range(B):
                torch.mm(wt, g[b].t(), out=dy[b])                  # (D, C) x (C, L)
        if ctx.needs_input_grad[0]:
            dw = torch.mm(g[0].t(), y[0].t())                      # (C, L) x (L, D)
            for b in range(1, B):
                dw.addmm_(g[b].t(), y[b].t())
            dw = dw.to(ctx.w_dtype)
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = g.sum(dim=(0, 1)).to(ctx.b_dtype)
        return dw, db, dy


def project_in(weight, u, bias=None):
    """in-projection of a (B, L, D) activation straight into the channels-first layout the convolutions read:
    (C, D) x (B, L, D) -> (B, C, L), one plain 2-D GEMM per batch row on the transposed view u[b].t(), written into its slice of
    the output (forward and backward: _ProjectIn).  The reference callers write `self.in_proj.weight @ u.transpose(-1, -2)`
    (hyenadna_flashfftconv.py:269-270, monarch_mixer_sequence_mixer_flashfftconv.py:124-125, bias dropped there too):
    torch.matmul folds that into a (B*L, D) x (D, C) GEMM and then copies the result into (B, C, L) with a strided elementwise
    kernel -- 224 of its 267 us at B2 L32K D256 on MI355X, 15 % of a HyenaDNA layer (benchmarks/scratch/proj_probe.py).
    (Not torch.bmm on the broadcast weight: on this ROCm 7.0 / PyTorch 2.10 stack the BATCHED GEMM with a transposed-view
    operand writes out of bounds at D = 768, L >= 8192 -- hipBLASLt and rocBLAS alike, the reference's own matmul form included;
    benchmarks/scratch/bmm_fault2.py.  The 2-D GEMM is the path every nn.Linear takes.)"""
    if u.shape[0] > _LOOP_MAX_BATCH:      # many short sequences: one GEMM + the layout copy, which is small there
        # (fp32 master weights under mixed precision: cast like _ProjectIn does; autograd returns the gradient in the weight's dtype)
        out = torch.nn.functional.linear(u, weight if weight.dtype == u.dtype else weight.to(u.dtype)).transpose(-1, -2).contiguous()
    else:
        out = _ProjectIn.apply(weight, u)
    return out if bias is None else out + bias.to(out.dtype).view(1, -1, 1)


def project_out(weight, bias, y):
    """out-projection of a channels-first (B, D, L) result back to (B, L, C): nn.Linear on y.transpose(-1, -2) first copies
    the transposed view (55 of 88 us at the shape above); a 2-D GEMM per batch row reads y[b].t() as its transposed operand."""
    if y.shape[0] > _LOOP_MAX_BATCH:
        w = weight if weight.dtype == y.dtype else weight.to(y.dtype)
        bb = bias if bias is None or bias.dtype == y.dtype else bias.to(y.dtype)
        return torch.nn.functional.linear(y.transpose(-1, -2), w, bb)
    return _ProjectOut.apply(weight, bias, y)


class FlashHyenaMixer(torch.nn.Module):
    """The whole order-2 Hyena operator of the reference callers (hyenadna_flashfftconv.py:228-289 HyenaOperator.forward):
    in_proj -> short depthwise conv -> x2 * fftconv(x1 * v, k) -> out_proj, on (B, L, D) activations, with no layout copy:
    the two projections are GEMMs on transposed views (project_in / project_out), the gates are read in place
    (FlashHyenaOp).  `in_proj` (D -> 3D) and `out_proj` (D -> D) are the caller's nn.Linear modules (shared, not copied);
    the in-projection bias is not applied, as in the reference (`self.in_proj.weight @ u`)."""

    def __init__(self, d_model, fft_size, in_proj, out_proj, short_filter_weight, short_filter_bias, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.in_proj, self.out_proj = in_proj, out_proj
        self.op = FlashHyenaOp(d_model, fft_size, short_filter_weight, short_filter_bias, dtype=dtype, device=device)

    def forward(self, u, k):
        y = self.op(project_in(self.in_proj.weight, u), k)
        return project_out(self.out_proj.weight, self.out_proj.bias, y)
