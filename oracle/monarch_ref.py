"""Structure-faithful float64 restatement of the REFERENCE's Monarch FFT decomposition
(TEST INFRASTRUCTURE).  Follows reference flashfftconv/conv.py:
  fft_matrix / ifft_matrix                         conv.py:22-26, 38-42
  compute_twiddle_factors_fft / _ifft              conv.py:28-36, 44-52
  2-stage k_f permutation                          conv.py:585
  3-stage k_f permutation (32x32x32)               conv.py:676
and the stage order of kernels_bf16/monarch_cuda_32_32_32_kernel_bf16.h:395-637.
It shows that the reference's factorisation computes exactly fft -> pointwise k_f -> ifft, i.e. the
same function as oracle/ref_fft_conv.py, which is what our (differently factorised) kernels are
tested against."""
import numpy as np


def fft_matrix(n):
    a = np.arange(n)
    return np.exp(-2j * np.pi * a[:, None] * a[None, :] / n)


def twiddle_fft(n, m):
    return np.exp(-2j * np.pi * np.arange(n)[:, None] * np.arange(m)[None, :] / (n * m))


def monarch_conv_2stage(u, k, N):
    """N = s*s (256, 1024): reference monarch_conv_forward path, conv.py:579-600."""
    s = int(round(np.sqrt(N)))
    F, Fi = fft_matrix(s), np.conj(fft_matrix(s))
    tw, twi = twiddle_fft(s, s) / N, np.conj(twiddle_fft(s, s))
    H = k.shape[0]
    k_f = np.fft.fft(k, n=N, axis=-1)
    k_f_perm = k_f.reshape(H, s, s).transpose(0, 2, 1).reshape(H, N)           # conv.py:585
    L = u.shape[-1]
    x = np.zeros(u.shape[:-1] + (N,)); x[..., :L] = u
    x = x.reshape(x.shape[:-1] + (s, s))
    x = np.einsum("ab,...bc->...ac", F, x) * tw          # DFT over the strided index, twiddle (carries 1/N)
    x = np.einsum("...ab,bc->...ac", x, F)               # DFT over the contiguous index
    x = x * k_f_perm.reshape(H, s, s)
    x = np.einsum("...ab,bc->...ac", x, Fi) * twi
    x = np.einsum("ab,...bc->...ac", Fi, x)
    return x.reshape(x.shape[:-2] + (N,)).real[..., :L]


def monarch_conv_3stage(u, k, N, n1, n2):
    """N = n1*n2*n2 (e.g. 32768 = 32*32*32): conv.py:672-691 + the kernel stage order."""
    H = k.shape[0]
    M = n2 * n2
    k_f = np.fft.fft(k, n=N, axis=-1)
    k_f_perm = (k_f.reshape(H, M, n1).transpose(0, 2, 1).reshape(H, n1, n2, n2).transpose(0, 1, 3, 2).reshape(H, N))  # conv.py:676
    F1, F2 = fft_matrix(n1), fft_matrix(n2)
    twN, tw2 = twiddle_fft(n1, M) / N, twiddle_fft(n2, n2)
    L = u.shape[-1]
    x = np.zeros(u.shape[:-1] + (N,)); x[..., :L] = u
    x = x.reshape(x.shape[:-1] + (n1, M))
    x = np.einsum("ab,...bc->...ac", F1, x) * twN                      # outer n1-point DFT + twiddle
    x = x.reshape(x.shape[:-1] + (n2, n2))
    x = np.einsum("ab,...bc->...ac", F2, x) * tw2
    x = np.einsum("...ab,bc->...ac", x, F2)
    x = x * k_f_perm.reshape(H, n1, n2, n2)
    x = np.einsum("...ab,bc->...ac", x, np.conj(F2)) * np.conj(tw2)
    x = np.einsum("ab,...bc->...ac", np.conj(F2), x)
    x = x.reshape(x.shape[:-2] + (M,)) * np.conj(twiddle_fft(n1, M))
    x = np.einsum("ab,...bc->...ac", np.conj(F1), x)
    return x.reshape(x.shape[:-2] + (N,)).real[..., :L]
