/* C-ABI of libflashfftconv_hip.so — the drop-in boundary for the FlashFFTConv hot path on MI355X.
 *
 * Replaces the pybind11 module `monarch_cuda` of the reference
 * (/root/reference/csrc/flashfftconv/monarch.cpp:14-59).  All pointers are DEVICE pointers owned
 * by the caller; every call only enqueues work on `stream` (a hipStream_t passed as void*).
 * Return value: 0 = ok, non-zero = error (message via ffc_last_error(), thread-local).
 * dtype: 0 = bf16, 1 = fp16 (activations, gates, k_f and outputs share it).
 */
#ifndef FLASHFFTCONV_HIP_H
#define FLASHFFTCONV_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffc_plan ffc_plan;

#define FFC_DT_BF16 0
#define FFC_DT_F16 1

int ffc_version(void);
const char* ffc_last_error(void);

/* Plan = DFT-tile operands + twiddle tables of one FFT size, uploaded once.
 * Replaces FlashFFTConv.__init__'s register_buffer tables (flashfftconv/conv.py:72-551). */
int ffc_plan_create(int64_t fft_size, int dtype, ffc_plan** out);
void ffc_plan_destroy(ffc_plan* plan);
/* Complex elements per head of k_f in the plan's internal ("Monarch") order. */
int64_t ffc_plan_kf_elems(const ffc_plan* plan);
/* internal position -> natural frequency index, ffc_plan_kf_elems() int32 entries (host memory). */
int ffc_plan_kf_index(const ffc_plan* plan, int32_t* out_host);
/* Scale folded into k_f (s_k) so that conv = iFFT(FFT(u) * FFT(k)) with no further factor. */
double ffc_plan_kf_scale(const ffc_plan* plan);

/* k (H, Lk) fp32  ->  k_f (H, kf_elems, 2) dtype in internal order, pre-scaled.
 * Replaces torch.fft.fft(k, n=N) + permute + cast (conv.py:572-575, :585, :676 ...). */
int ffc_kernel_fft(const ffc_plan* plan, const float* k, int64_t H, int64_t Lk, void* kf_out, void* stream);
/* Same from a natural-order complex64 spectrum (H, N) (e.g. an externally computed FFT). */
int ffc_kf_pack(const ffc_plan* plan, const void* kf_natural_c64, int64_t H, void* kf_out, void* stream);

/* y[b,h,:L] = postgate * iFFT(FFT(u*pregate, N) * k_f[h]).real[:L]      (gates nullable, both or none
 * not required here).  conj_kf=1 multiplies by conj(k_f): the input-gradient pass.
 * Covers monarch_conv_forward{,_r2r,_16_16_16,_32_16_16,_16_32_32,_32_32_32} (monarch.cpp:16-21)
 * and the dx half of the matching *_backward exports. */
int ffc_conv_fwd(const ffc_plan* plan, const void* u, const void* kf, const void* pregate, const void* postgate,
                 void* y, int64_t B, int64_t H, int64_t L, int conj_kf, void* stream);

/* dk_f accumulation: dkf[h, :] (fp32 complex, internal order, scaled) = sum_b FFT(dout*postgate) * conj(FFT(u*pregate)).
 * ws: workspace of ffc_dkf_workspace_bytes() bytes (partial sums per chunk of batch pairs). */
int64_t ffc_dkf_workspace_bytes(const ffc_plan* plan, int64_t B, int64_t H);
int ffc_conv_bwd_dkf(const ffc_plan* plan, const void* dout, const void* u, const void* pregate, const void* postgate,
                     void* ws, int64_t B, int64_t H, int64_t L, void* stream);
/* dk (H, Lk) fp32 = real(iFFT(sum of partials))[:Lk].  Replaces dk_f_out.sum(0) + un-permute +
 * torch.fft.ifft(..., norm='forward').real[..., :k_len] (conv.py:1758-1761, 1861-1864). */
int ffc_kernel_ifft_grad(const ffc_plan* plan, const void* ws, int64_t B, int64_t H, int64_t Lk, float* dk, void* stream);

/* Short depthwise conv1d (reference csrc/flashfftconv/conv1d/conv1d.h:48-95).
 * in_dtype / w_dtype: 0 bf16, 1 fp16, 2 fp32.  is_bhl: u (B,D,L) w (D,K) else u (B,L,D) w (K,D). */
int ffc_conv1d_fwd(const void* u, const void* w, const void* bias, void* y, int in_dtype, int w_dtype,
                   int64_t B, int64_t D, int64_t L, int K, int P, int is_bhl, void* stream);
int ffc_conv1d_bwd(const void* dout, const void* u, const void* w, void* du, float* dw, float* dbias,
                   int in_dtype, int w_dtype, int64_t B, int64_t D, int64_t L, int K, int P, int is_bhl,
                   void* stream);

/* Hardware-primitive self test (MFMA lane layouts, ds_read_b64_tr_b16, packing): fills `out_host`
 * (host memory, 64*40 uint32) for comparison against the CPU wave simulator.  Test support. */
int ffc_selftest_primitives(const uint32_t* in_host, uint32_t* out_host);

#ifdef __cplusplus
}
#endif
#endif
