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
/* Tuning knobs (FFC_FLAGS, FFC_STREAM, FFC_PERSIST, FFC_WG_MULT) are read from the environment once, by ffc_plan_create;
 * this re-reads them into an existing plan (A/B tuning scripts).  No launch path calls getenv. */
void ffc_plan_reload_env(ffc_plan* plan);
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

/* Same with batch strides in ELEMENTS (0 = contiguous, H * L): row (b, h) of a tensor starts at b * stride + h * L, so u,
 * the gates and y may each be a channel slice of a wider (B, C, L) tensor, read / written in place.  This is what lets the
 * fused Hyena / M2 operator (flashfftconv/hyena.py; reference callers examples/hyena-dna/hyenadna_flashfftconv.py:274-284,
 * examples/bert/monarch_mixer_sequence_mixer_flashfftconv.py:127-170) feed x1, x2, v = split(short_conv(...)) to the
 * gated kernel without the x1*v, y*x2 elementwise kernels and the .contiguous() copies. */
int ffc_conv_fwd_strided(const ffc_plan* plan, const void* u, const void* kf, const void* pregate, const void* postgate,
                         void* y, int64_t B, int64_t H, int64_t L, int conj_kf, int64_t sb_u, int64_t sb_pre,
                         int64_t sb_post, int64_t sb_y, void* stream);

/* ffc_conv_fwd for a LOW-PASS k_f: every non-zero bin has k3 = f / (N1 N2) < rows or >= 32 - rows, rows <= 4, i.e.
 * |f| < rows * N / 32 (FrequencySparseFFTConv with N_partial <= N / 4; the caller still passes the masked k_f).  Same result;
 * the kernel skips the all-zero spectrum rows (half of the k_f loads and product, one of two K-steps of the first inverse stage).
 * fft 16384 / 32768.  Reference: the truncated kernels, csrc/flashfftconv/monarch_cuda/monarch_fwd_complex.h:462-528. */
int ffc_conv_fwd_sparse(const ffc_plan* plan, const void* u, const void* kf, const void* pregate, const void* postgate, void* y,
                        int64_t B, int64_t H, int64_t L, int conj_kf, int rows, void* stream);

/* dk_f accumulation: dkf[h, :] (fp32 complex, internal order, scaled) = sum_b FFT(dout*postgate) * conj(FFT(u*pregate)).
 * ws: workspace of ffc_dkf_workspace_bytes() bytes (partial sums per chunk of batch pairs). */
int64_t ffc_dkf_workspace_bytes(const ffc_plan* plan, int64_t B, int64_t H);
/* The workspace starts with ffc_dkf_slab_count() fp32 slabs [slab][H][kf_elems][2] in k_f's internal order
 * (frequency-sparse convolutions mask them before ffc_kernel_ifft_grad; reference flashfftconv/sparse_conv.py:24-38). */
int64_t ffc_dkf_slab_count(const ffc_plan* plan, int64_t B, int64_t H);
int ffc_conv_bwd_dkf(const ffc_plan* plan, const void* dout, const void* u, const void* pregate, const void* postgate,
                     void* ws, int64_t B, int64_t H, int64_t L, void* stream);
/* Fused backward (one launch): du = pregate * corr(dout*postgate, k), dpre = u * corr(...) (nullable) and the
 * dk_f partial sums in ws.  Covers monarch_conv_backward{,_r2r,_16_16_16,...} (monarch.cpp:27-32): three
 * transforms per pair like the reference's bwd kernels, dk_f kept in fp32. */
int ffc_conv_bwd(const ffc_plan* plan, const void* dout, const void* u, const void* kf, const void* pregate,
                 const void* postgate, void* du, void* dpre, void* ws, int64_t B, int64_t H, int64_t L, void* stream);
/* Same plus dpost = dout * conv(u*pregate, k) (nullable): the gated backward of GatedFlashFFTConvFunc
 * (conv.py:3236-4958) in one call.  For fft sizes >= 4096 dpost comes out of the same launch (the forward output is one
 * extra inverse transform of the first spectrum of each pair); the reference recomputes the forward for it too. */
int ffc_conv_bwd_gated(const ffc_plan* plan, const void* dout, const void* u, const void* kf, const void* pregate,
                       const void* postgate, void* du, void* dpre, void* dpost, void* ws, int64_t B, int64_t H, int64_t L,
                       void* stream);
int ffc_conv_bwd_gated_strided(const ffc_plan* plan, const void* dout, const void* u, const void* kf, const void* pregate,
                               const void* postgate, void* du, void* dpre, void* dpost, void* ws, int64_t B, int64_t H,
                               int64_t L, int64_t sb_dout, int64_t sb_u, int64_t sb_pre, int64_t sb_post, int64_t sb_du,
                               int64_t sb_dpre, int64_t sb_dpost, void* stream);
/* Spectrum-saving pair of the two calls above (MI355X design, no reference counterpart: the reference's backward kernels
 * recompute FFT(u * pregate), kernels_bf16/monarch_cuda_32_32_32_bwd_kernel_bf16.h).  ffc_conv_fwd_z also stores every batch
 * pair's spectrum FFT(u * pregate) (plan dtype, internal order, ffc_spectrum_bytes() bytes: 2x the size of u at L = N/2) and
 * ffc_conv_bwd_z reads it instead of transforming u again: one of the backward's three transforms per pair, its rows of u
 * and its scratch round trip disappear.  du / dpregate come out bit for bit as from ffc_conv_bwd_gated_strided; dk and dpostgate
 * agree to the rounding of the spectrum (the forward and the backward kernel schedule the same fp32 operations differently).
 * Every fused plan has the pair (fft 256 ... 131072); layout [H][pair][fft size] complex values, for the single-tile sizes
 * (fft <= 2048) one 4 KB slot per tile of G pairs and pass.  There the forward output of ffc_conv_fwd_z agrees with
 * ffc_conv_fwd to last-bit steps of the dtype (two instantiations of the same arithmetic), for fft >= 4096 bit for bit.
 * y_raw (nullable; contiguous (B,H,L) dtype): the forward output before the postgate multiply.  The gated backward's
 * dpostgate is dout * y_raw -- with it the caller passes dpost = NULL to ffc_conv_bwd_z and that kernel runs no third transform.
 * Single-tile sizes (fft <= 2048), round 6: y_raw may be kept WITHOUT the spectra -- ffc_conv_fwd_z / ffc_conv_fwd_k with zsave = NULL and
 * y_raw given, ffc_conv_bwd_zy / ffc_conv_bwd_k with zin = NULL and y_raw given: the backward transforms u * pregate again (bit for bit the
 * recomputing kernel's du, dpregate and dk_f sums) and takes dpostgate from y_raw.  A third less kept memory; the module's FFC_Y_ONLY_MAX. */
int64_t ffc_spectrum_bytes(const ffc_plan* plan, int64_t B, int64_t H);
int ffc_conv_fwd_z(const ffc_plan* plan, const void* u, const void* kf, const void* pregate, const void* postgate, void* y,
                   void* zsave, void* y_raw, int64_t B, int64_t H, int64_t L, int64_t sb_u, int64_t sb_pre, int64_t sb_post,
                   int64_t sb_y, void* stream);
int ffc_conv_bwd_z(const ffc_plan* plan, const void* dout, const void* u, const void* kf, const void* pregate,
                   const void* postgate, void* du, void* dpre, void* dpost, void* ws, const void* zin, int64_t B, int64_t H,
                   int64_t L, int64_t sb_dout, int64_t sb_u, int64_t sb_pre, int64_t sb_post, int64_t sb_du, int64_t sb_dpre,
                   int64_t sb_dpost, void* stream);
/* ffc_conv_bwd_z with the y_raw that ffc_conv_fwd_z stored: dpost = dout * y_raw (fp32 product, rounded once: the same
 * arithmetic as the forward's output gate) is written while the kernel loads the rows of dout -- all five gradients of the gated
 * backward from one launch, as reference flashfftconv/conv.py:3979 returns them from native code; no third transform. */
int ffc_conv_bwd_zy(const ffc_plan* plan, const void* dout, const void* u, const void* kf, const void* pregate,
                    const void* postgate, void* du, void* dpre, void* dpost, void* ws, const void* zin, const void* y_raw,
                    int64_t B, int64_t H, int64_t L, int64_t sb_dout, int64_t sb_u, int64_t sb_pre, int64_t sb_post,
                    int64_t sb_du, int64_t sb_dpre, int64_t sb_dpost, void* stream);
/* One call per direction of the module (MI355X host path: at short sequences a training step is bound by the host, and every trip
 * through the binding costs 4-8 us).  ffc_conv_fwd_k = ffc_kernel_fft(k -> kf_out) + ffc_conv_fwd / ffc_conv_fwd_z (zsave, y_raw
 * nullable) -- what reference FlashFFTConvFunc.forward does in one Python function, conv.py:572-588.  ffc_conv_bwd_k = the fused
 * backward (on zin / y_raw when given, recomputing otherwise) + ffc_kernel_ifft_grad: all five gradients of conv.py:1737-1761 /
 * :3979 from one call.  Contiguous tensors; same kernels and results as the separate entry points. */
int ffc_conv_fwd_k(const ffc_plan* plan, const float* k, int64_t Lk, void* kf_out, const void* u, const void* pregate,
                   const void* postgate, void* y, void* zsave, void* y_raw, int64_t B, int64_t H, int64_t L, void* stream);
int ffc_conv_bwd_k(const ffc_plan* plan, const void* dout, const void* u, const void* kf, const void* pregate, const void* postgate,
                   void* du, void* dpre, void* dpost, void* ws, const void* zin, const void* y_raw, float* dk, int64_t Lk, int64_t B,
                   int64_t H, int64_t L, void* stream);
/* The same pair for the INNER convolution of the HBM-level sizes (fft >= 262144; flashfftconv/bigfft.py): ffc_conv_fwd_kx =
 * ffc_kernel_fft_c (complex rows -> kf_out, `scale`) + the ungated forward (spectra kept in zsave when given); ffc_conv_bwd_kx = the
 * fused backward + ffc_kernel_ifft_grad_c (dk rows as a complex pair-plane tensor (2, H, N) bf16, `scale`).  One launch each where
 * a workgroup owns its row. */
int ffc_conv_fwd_kx(const ffc_plan* plan, const void* xpair, float scale, void* kf_out, const void* u, void* y, void* zsave,
                    int64_t B, int64_t H, int64_t L, void* stream);
int ffc_conv_bwd_kx(const ffc_plan* plan, const void* dout, const void* u, const void* kf, void* du, void* ws, const void* zin,
                    void* outpair, float scale, int64_t B, int64_t H, int64_t L, void* stream);
/* dk (H, Lk) fp32 = real(iFFT(sum of partials))[:Lk].  Replaces dk_f_out.sum(0) + un-permute +
 * torch.fft.ifft(..., norm='forward').real[..., :k_len] (conv.py:1758-1761, 1861-1864). */
int ffc_kernel_ifft_grad(const ffc_plan* plan, const void* ws, int64_t B, int64_t H, int64_t Lk, float* dk, void* stream);
/* Same from `nslab` caller-owned fp32 slabs [nslab][H][kf_elems][2]: callers that reduce the partial sums themselves
 * (multi-GPU B-shard, flashfftconv/sharding.py: reduce-scatter of dk_f over RCCL, then H/W heads per rank). */
int ffc_kernel_ifft_grad_slabs(const ffc_plan* plan, const void* slabs, int64_t nslab, int64_t H, int64_t Lk, float* dk,
                               void* stream);

/* FFT sizes 65536 and 131072 are PLAN sizes too (ffc_plan_create(65536 | 131072)): they run as 2 / 4 passes of the fused 32768
 * kernel over the same rows (csrc/ffc_body.h struct Pass) through the entry points above; k_f then holds R * 32 tiles per head.
 * FFT sizes 262144..4194304 (and 65536 / 131072 on request) = one or two outer DFT levels (factor n0 = 16 or 32) through HBM around a
 * fused inner size (replaces butterfly_{,padded_}{,gated_}{,ifft_}*forward, monarch.cpp:41-56, and the
 * *_complex monarch exports :22-38).  The host chains the passes (flashfftconv/bigfft.py), exactly as
 * reference conv.py:1420-1524 chains butterfly -> inner -> butterfly_ifft.
 * dir=1: in (Bv,Hin,Llong) real/pair rows [* gate] -> out (2*npair, Hin*n0, Mi) pair-plane complex.
 * dir=0: the inverse map, [* gate] applied to the output.  plan16/plan32: any plans whose outer digit
 * is 16 / 32 (e.g. fft sizes 16384 / 32768), they supply the DFT tile in `dtype`. */
/* `dtype` of the three level entry points: 0 bf16 / 1 fp16, optionally | 16: the LONG side is fp32 -- dir = 1: `in` is float, multiplied
 * by 2^e (e = bits 8..15 of dtype, the fp16 mode's prescale of the filter) and rounded once to the 16-bit type in the row load;
 * dir = 0: `out` is float (the 16-bit results widened).  The filter k and its gradient dk pass through the levels without cast
 * kernels this way.  Gates stay 16-bit.
 * | 32 (round 6, "half rows"): the long side is ONE REAL row per head (Bv == 1, npair == 1: a batch of one, the filter k, dk).  Its level
 * rows are conjugate mirrors, x_{K-k0}[m] = W_Mi^m conj(x_k0[m]) with K = n0 (or R * 32) rows per head, so the short side holds the
 * K / 2 + 1 rows k0 <= K / 2 only -- (2, Hin * (K / 2 + 1), Mi), pass c of an R-pass factor first its d <= 16 (c = 0) resp. d < 16 rows --
 * and dir = 0 rebuilds the real output from them (weights 1, 2, .., 2, 1).  Half of the inner convolution's rows: what the reference's
 * r2c / c2r kernels save at B = 1 (csrc/flashfftconv/monarch_cuda/kernels_bf16/monarch_cuda_shared_r2r_bf16.h:92-239). */
int ffc_outer_pass(const ffc_plan* plan16, const ffc_plan* plan32, int n0, int dtype, int dir, const void* in, void* out,
                   const void* gate, int64_t Bv, int64_t npair, int64_t Hin, int64_t Mi, int64_t Llong, float scale,
                   void* stream);
/* complex-input k_f and complex-output dk variants used by the big sizes (pair-plane tensors (2,H,N)). */
/* One level of factor R * 32 as R passes c = 0 .. R-1 of the 32-point kernel (fft 4194304 = 128 x 32768 in ONE level when the
 * long side is at most N / R: plan_r = the multi-pass plan with that R, fft 131072 for R = 4).  dir = 0: call the passes in
 * order (c > 0 adds its -- gated -- contribution to the long side). */
int ffc_outer_pass_r(const ffc_plan* plan_r, int c, int dtype, int dir, const void* in, void* out, const void* gate, int64_t Bv,
                     int64_t npair, int64_t Hin, int64_t Mi, int64_t Llong, float scale, void* stream);
/* The same level as the R calls ffc_outer_pass_r(plan_r, c = 0 .. R-1, ...) in ONE launch: the forward reads the long side once
 * (not R times), the inverse sums the passes in fp32 and writes the long side once (no read-modify-write between launches).
 * Round 6: Llong may exceed 32 * Mi (up to R * 32 * Mi, i.e. any length of the level: the reference's 128-point butterfly takes any length too,
 * csrc/flashfftconv/butterfly/butterfly_padded_cuda_bf16.cu:302-487) -- the WIDE form: an R-point butterfly of the R long-side row blocks in front of
 * the pass matrices.  16-bit long side only (no | 16, no | 32 in `dtype`).  Slower than two levels on MI355X (profiles/r06_ab_wide.txt) and a quarter
 * less peak memory: the module takes it on request (FFC_BIG_WIDE=1). */
int ffc_outer_pass_all(const ffc_plan* plan_r, int dtype, int dir, const void* in, void* out, const void* gate, int64_t Bv,
                       int64_t npair, int64_t Hin, int64_t Mi, int64_t Llong, float scale, void* stream);
int ffc_kernel_fft_c(const ffc_plan* plan, const void* xpair, int64_t H, void* kf_out, float scale, void* stream);
int ffc_kernel_ifft_grad_c(const ffc_plan* plan, const void* ws, int64_t B, int64_t H, void* outpair, float scale, void* stream);
/* ... from `nslab` caller-owned fp32 slabs [nslab][H][kf_elems][2] (multi-GPU B-shard: rows reduce-scattered over the ranks) */
int ffc_kernel_ifft_grad_c_slabs(const ffc_plan* plan, const void* slabs, int64_t nslab, int64_t H, void* outpair, float scale,
                                 void* stream);

/* Short depthwise conv1d (reference csrc/flashfftconv/conv1d/conv1d.h:48-95).
 * in_dtype / w_dtype: 0 bf16, 1 fp16, 2 fp32.  is_bhl: u (B,D,L) w (D,K) else u (B,L,D) w (K,D). */
int ffc_conv1d_fwd(const void* u, const void* w, const void* bias, void* y, int in_dtype, int w_dtype,
                   int64_t B, int64_t D, int64_t L, int K, int P, int is_bhl, void* stream);
int ffc_conv1d_bwd(const void* dout, const void* u, const void* w, void* du, float* dw, float* dbias,
                   int in_dtype, int w_dtype, int64_t B, int64_t D, int64_t L, int K, int P, int is_bhl,
                   void* stream);

/* Profiling build of the N=32768 bf16 forward kernel: per-wave cycle sums of its phases
 * (rows-in, phase A, barrier, phase B, barrier, phase C, rows-out) in prof[grid][8 waves][8].  Tuning support. */
int ffc_conv_fwd_prof(const ffc_plan* plan, const void* u, const void* kf, void* y, int64_t B, int64_t H, int64_t L,
                      unsigned long long* prof, int* grid_out, void* stream);

/* Test support: fills every CU's LDS and vector/accumulation registers with NaN patterns (a kernel that reads
 * state it never initialised then fails loudly instead of inheriting the previous workgroup's benign leftovers). */
int ffc_debug_poison(void* stream);
/* measured peaks of the current device: stream copy (GB/s, read + write bytes) and dense v_mfma_f32_32x32x16_bf16 (TFLOP/s) */
int ffc_debug_peaks(double* copy_GBs, double* mfma_TFLOPs);

/* Hardware-primitive self test (MFMA lane layouts, ds_read_b64_tr_b16, packing): fills `out_host`
 * (host memory, 64*40 uint32) for comparison against the CPU wave simulator.  Test support. */
int ffc_selftest_primitives(const uint32_t* in_host, uint32_t* out_host);

#ifdef __cplusplus
}
#endif
#endif
