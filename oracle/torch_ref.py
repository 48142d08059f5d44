"""Torch restatement of the reference's CPU-runnable oracle (TEST INFRASTRUCTURE / cpu_baseline only).
  ref_fft_conv : /root/reference/tests/test_flashfftconv.py:5-13 (also benchmarks/benchmark_flashfftconv.py:10-16)
Gradients come from autograd through it, exactly like the reference tests (test_flashfftconv.py:85-107)."""
import torch


def ref_fft_conv(u, k, n=None):
    if n is None:
        n = u.size(-1)
    l = u.size(-1)
    u_f = torch.fft.fft(u.to(torch.float32), n=n)
    k_f = torch.fft.fft(k.to(torch.float32), n=n)
    out = torch.fft.ifft(u_f * k_f, n=n)
    return out.real.to(u.dtype)[..., :l]


def ref_fft_conv_gated(u, k, pregate, postgate, n=None):
    return ref_fft_conv(u * pregate, k, n) * postgate
