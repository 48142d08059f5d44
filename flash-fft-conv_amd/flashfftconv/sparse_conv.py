"""Partial and frequency-sparse convolutions on the MI355X FlashFFTConv kernels.

Same classes, constructor arguments and semantics as the reference's torch.fft demos
(/root/reference/flashfftconv/sparse_conv.py:8-38), but running through the fused HIP path:

  PartialFFTConv(N_partial)(x, k)          y = conv(x, k[..., :N_partial]),             fft size 2L
  FrequencySparseFFTConv(N_partial)(x, k)  y = irfft(rfft(x) * rfft(k) * [f < N_partial // 2]), fft size 2L

x is (B, H, L) bf16/fp16 on the GPU, k is (H, Lk) fp32 (the reference demos up-cast x to fp32 and back;
here the module dtype is x's dtype).  Both are differentiable in x and k.  The frequency-sparse variant
zeroes k_f's bins in the library's internal order (mask built once from ffc_plan_kf_index) and masks the
fp32 dk_f partial sums the same way in the backward, at every fft size the module supports (sequence lengths up to 2097152: above
fft 131072 the mask goes onto the inner k_f / dk_f rows of the HBM-level form, flashfftconv/bigfft.py row_freq).
"""
import torch

from .conv import FlashFFTConv


class _ConvCache(torch.nn.Module):
    def __init__(self, N_partial):
        super().__init__()
        self.N_partial = N_partial
        self._convs = {}

    def _conv_for(self, x, keep=None):
        key = (2 * x.shape[-1], x.dtype)
        conv = self._convs.get(key)
        if conv is None:
            conv = FlashFFTConv(key[0], dtype=x.dtype)
            if keep is not None:
                # (fft sizes with HBM levels, >= 262144: the mask is applied to the inner k_f / dk_f rows, conv._big_kf_mask)
                # folded size (2048 on the 4096 plan): natural bin f of the 2048-point spectrum is bin 2f of the plan
                conv._kf_keep = keep * (conv._plan_seqlen // conv.seqlen)
            self._convs[key] = conv
        conv.train(self.training)
        return conv


class PartialFFTConv(_ConvCache):
    """reference flashfftconv/sparse_conv.py:8-22"""

    def forward(self, x, k):
        return self._conv_for(x)(x, k[..., : self.N_partial])


class FrequencySparseFFTConv(_ConvCache):
    """reference flashfftconv/sparse_conv.py:24-38: rfft bins >= N_partial // 2 of k are zeroed."""

    def forward(self, x, k):
        return self._conv_for(x, keep=self.N_partial // 2)(x, k)
