"""FlashDepthWiseConv1d for MI355X (drop-in for reference flashfftconv/depthwise_1d.py:7-55)."""
import os
import torch

from . import _lib

# The reference's BLH backward returns the weight gradient with the WRONG memory layout: it computes dk as (d, k) and hands it
# back as `.view({k, d})` -- a reinterpretation, not a transpose (csrc/flashfftconv/conv1d/conv1d_bwd_cuda_blh.cu:115) -- so
# `weights.grad` (k, d) of an is_bhl=False module holds the (d, k)-ordered numbers, and its own test compares
# `weights.grad.view(d, k)` (tests/test_conv1d.py:220).  This package returns the gradient OF the (k, d) parameter.
# FFC_REF_BLH_GRAD_LAYOUT=1 reproduces the reference's layout (tests/test_reference_verbatim_gpu.py sets it to run the
# reference's test file unmodified).
_REF_BLH_GRAD_LAYOUT = os.environ.get("FFC_REF_BLH_GRAD_LAYOUT", "0") == "1"

_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}


class _Conv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weights, bias, padding, is_bhl):
        if not x.is_cuda:
            raise RuntimeError("FlashDepthWiseConv1d: input must be a CUDA/HIP tensor")
        x = x.contiguous()
        weights = weights.contiguous()
        bias = bias.contiguous()
        if is_bhl:
            B, D, L = x.shape
            K = weights.shape[1]
        else:
            B, L, D = x.shape
            K = weights.shape[0]
        if K % 2 != 1:
            raise RuntimeError("FlashDepthWiseConv1d: kernel size must be odd")   # conv1d/conv1d.h:68
        Lout = L + 2 * padding - K + 1
        y = torch.empty((B, D, Lout) if is_bhl else (B, Lout, D), dtype=x.dtype, device=x.device)
        bias_w = bias.to(weights.dtype)
        _lib.check(_lib.lib().ffc_conv1d_fwd(_lib.ptr(x), _lib.ptr(weights), _lib.ptr(bias_w), _lib.ptr(y), _DT[x.dtype],
                                             _DT[weights.dtype], B, D, L, K, padding, int(is_bhl), _lib.stream_ptr()),
                   "ffc_conv1d_fwd")
        ctx.save_for_backward(x, weights, bias)
        ctx.padding, ctx.is_bhl, ctx.dims = padding, is_bhl, (B, D, L, K)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, weights, bias = ctx.saved_tensors
        B, D, L, K = ctx.dims
        dout = dout.contiguous()
        du = torch.empty_like(x)
        dw = torch.zeros(weights.shape, dtype=torch.float32, device=x.device)
        db = torch.zeros(D, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ffc_conv1d_bwd(_lib.ptr(dout), _lib.ptr(x), _lib.ptr(weights), _lib.ptr(du), _lib.ptr(dw),
                                             _lib.ptr(db), _DT[x.dtype], _DT[weights.dtype], B, D, L, K, ctx.padding,
                                             int(ctx.is_bhl), _lib.stream_ptr()), "ffc_conv1d_bwd")
        if _REF_BLH_GRAD_LAYOUT and not ctx.is_bhl:
            dw = dw.t().contiguous().view(K, D)
        return du, dw.to(weights.dtype), db.to(bias.dtype), None, None


class FlashDepthWiseConv1d(torch.nn.Module):
    def __init__(self, channels, kernel_size, padding, weights, bias, is_bhl=True, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d = channels
        self.k = kernel_size
        self.padding = padding
        self.is_bhl = is_bhl
        w = weights.detach().clone().squeeze()
        if w.dim() == 1:
            w = w.view(channels, kernel_size)
        if not is_bhl:
            w = w.transpose(0, 1).contiguous()
        self.weights = torch.nn.Parameter(w.to(**{k: v for k, v in factory_kwargs.items() if v is not None}))
        self.bias = torch.nn.Parameter(bias.detach().clone().to(**{k: v for k, v in factory_kwargs.items() if v is not None}))

    def forward(self, input):
        return _Conv1dFn.apply(input, self.weights, self.bias, self.padding, self.is_bhl)
