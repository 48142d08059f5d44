// Fused backward kernel (Modes::bwd) + ffc_conv_bwd / ffc_conv_bwd_gated(_strided).  (Split from ffc_k_dkf.hip so that the two
// longest translation units of the library compile side by side.)
#include "ffc_bwd_launch.h"

// out[b, h, n] = x[b, h, n] * y[b, h, n] on (B, H, L) rows with batch strides (elements), fp32 product rounded once: the
// dpostgate = dout * y_raw of the multi-pass sizes (fft 65536 / 131072), whose backward kernel has no registers to spare for it
template <int DT>
__global__ __launch_bounds__(256) void mul_rows_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ y, uint16_t* __restrict__ out,
                                                       int64_t HL, int64_t sbx, int64_t sby, int64_t sbo, int fast) {
  const int b = blockIdx.y;
  const uint16_t* xb = x + b * sbx; const uint16_t* yb = y + b * sby; uint16_t* ob = out + b * sbo;
  if (fast) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HL / 8; i += (int64_t)gridDim.x * blockDim.x) {
      const u32x4v a = __builtin_nontemporal_load((const u32x4v*)xb + i), c = __builtin_nontemporal_load((const u32x4v*)yb + i);
      u32x4v o;
#pragma unroll
      for (int q = 0; q < 4; q++)
        o[q] = DevB::pack<DT>(DevB::unpack_lo<DT>(a[q]) * DevB::unpack_lo<DT>(c[q]), DevB::unpack_hi<DT>(a[q]) * DevB::unpack_hi<DT>(c[q]));
      __builtin_nontemporal_store(o, (u32x4v*)ob + i);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HL; i += (int64_t)gridDim.x * blockDim.x)
      ob[i] = (uint16_t)DevB::pack<DT>(DevB::unpack_lo<DT>(xb[i]) * DevB::unpack_lo<DT>(yb[i]), 0.f);
  }
}


#if defined(FFC_BWD_PROF)
// profiling variant only (build.py --variant bwdprof -DFFC_BWD_PROF): per-phase s_memtime sums, [workgroup][wave][16]
static unsigned long long* ffc_bwd_prof_buffer() {
  static unsigned long long* buf = nullptr;
  if (!buf && hipMalloc((void**)&buf, 8192 * 8 * 16 * 8) != hipSuccess) buf = nullptr;
  return buf;
}
extern "C" int ffc_debug_bwd_prof(unsigned long long* out_host, int64_t n_words) {
  unsigned long long* b = ffc_bwd_prof_buffer();
  if (!b || n_words > 8192 * 8 * 16) return ffc_fail("no profile buffer");
  return hipMemcpy(out_host, b, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : ffc_fail("copy failed");
}
#endif

// Fused backward: du = pregate * corr(dout*postgate, k), dpre = u * corr(...) (nullable, gated only) and the
// dk_f partial sums in `ws` (same layout as ffc_conv_bwd_dkf; finish with ffc_kernel_ifft_grad).
extern "C" int ffc_conv_fwd(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate, void* y,
                            int64_t B, int64_t H, int64_t L, int conj_kf, void* stream);
extern "C" int ffc_conv_bwd_gated(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                                  const void* postgate, void* du, void* dpre, void* dpost, void* ws, int64_t B, int64_t H,
                                  int64_t L, void* stream);
extern "C" int ffc_conv_bwd(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                            const void* postgate, void* du, void* dpre, void* ws, int64_t B, int64_t H, int64_t L, void* stream) {
  return ffc_conv_bwd_gated(p, dout, u, kf, pregate, postgate, du, dpre, nullptr, ws, B, H, L, stream);
}
// + dpost = dout * conv(u*pregate, k) (nullable).  Fused sizes >= 4096 produce it inside the same launch (one extra
// inverse transform per pair); the single-tile sizes (fft <= 2048: the 1024 kernel and its 2-pass form) run the forward kernel with dout as the output gate.
// Batch strides in elements (0 = contiguous H * L): every tensor may be a channel slice of a wider (B, C, L) tensor.
extern "C" int ffc_conv_fwd_strided(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate,
                                    void* y, int64_t B, int64_t H, int64_t L, int conj_kf, int64_t sb_u, int64_t sb_pre,
                                    int64_t sb_post, int64_t sb_y, void* stream);
extern "C" int64_t ffc_spectrum_bytes(const ffc_plan* p, int64_t B, int64_t H);
static int conv_bwd_impl(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                         const void* postgate, void* du, void* dpre, void* dpost, void* ws, const void* zin, int64_t B, int64_t H,
                         int64_t L, int64_t sb_dout, int64_t sb_u, int64_t sb_pre, int64_t sb_post,
                         int64_t sb_du, int64_t sb_dpre, int64_t sb_dpost, void* stream, const void* yraw = nullptr,
                         float* dk = nullptr, int64_t Lk = 0, bool* dk_done = nullptr, void* dk_pair = nullptr, float dk_pair_scale = 1.0f) {
  if (!p || !dout || !kf || !du || !ws) return ffc_fail("null arg");
  // `u` is read by the recomputing kernels, as the gate of dpregate and by the forward run that makes dpostgate without y_raw; on saved
  // spectra without gates nothing reads it and a null pointer says so (the HBM-level sizes' inner call, flashfftconv/conv.py _big_backward)
  if (!u && (!zin || pregate || postgate || dpre || (dpost && !yraw))) return ffc_fail("null u: only with saved spectra and no gates");
  if (yraw && !dpost) return ffc_fail("y_raw needs a dpost output");
  // (single-tile sizes, fft <= 2048: y_raw also without the spectra -- the kernel transforms u * pregate again, rows it loads anyway)
  if (yraw && !zin && p->hp.N1 > 1) return ffc_fail("y_raw without the saved spectra: single-tile sizes (fft <= 2048) only");
  if (zin && (ffc_spectrum_bytes(p, B, H) == 0 || ((uintptr_t)zin & 15))) return ffc_fail("spectrum buffer: unsupported plan or misaligned");
  if (B <= 0 || H <= 0) return ffc_fail("empty batch/heads");
  if (L <= 0 || L > p->hp.N) return ffc_fail("L must be in (0, fft_size]");
  if ((uintptr_t)kf & 15) return ffc_fail("k_f must be 16-byte aligned");
  int64_t* sbs[7] = {&sb_dout, &sb_u, &sb_pre, &sb_post, &sb_du, &sb_dpre, &sb_dpost};
  int64_t any = 0;
  for (int i = 0; i < 7; i++) {
    if (*sbs[i] == 0) *sbs[i] = H * L;
    if (*sbs[i] < H * L || (B - 1) * *sbs[i] + H * L >= ((int64_t)1 << 31))
      return ffc_fail("tensor too large (>= 2^31 elements) or batch stride smaller than H*L");
    any |= *sbs[i];
  }
  DkfArgs d{};
  ConvArgs& a = d.c;
  a.u = u; a.kf = kf; a.pregate = pregate; a.postgate = postgate; a.tab = p->d_blob; a.t = p->hp.tabs;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.npair = (int)((B + 1) / 2); a.s_inv = (float)p->hp.s_inv; a.s_fwd = (float)p->hp.s_fwd;
  a.sbu = sb_u; a.sbg = sb_pre; a.sbp = sb_post; a.sby = sb_du;
  d.sbd = sb_dout; d.sbdu = sb_du; d.sbdpre = sb_dpre; d.sbdpost = sb_dpost;
  a.fast = (L % 8 == 0) && !(((uintptr_t)u | (uintptr_t)dout | (uintptr_t)pregate | (uintptr_t)postgate | (uintptr_t)du | (uintptr_t)dpre) & 15) &&
           !(any & 7);
  ffc_choose_chunks(p, a.H, a.npair, &a.nchunk, &a.ppc);
  a.persist = ffc_persist(p);
  a.R = p->hp.R;
  a.stream = p->env_stream >= 0 ? p->env_stream : ((!pregate && !postgate && p->hp.R == 1) ? 1 : 0);    // see Body::STREAM_ROWS
  a.flags = p->env_flags;                        // tuning flags: 2 = k_f streamed, 4 = scratch streamed
  d.dout = dout; d.ws = (float*)ws; d.du = du; d.dpre = dpre; d.zscratch = ffc_zscratch(p, ws, a.H, a.nchunk);
  // dk from the same launch (Modes::dk_tail / dk_tail_multi): the workgroup owns all pairs of its head, fft 16384 / 32768 (8192 on request)
  // (tuning flag 32 keeps the slab + ffc_kernel_ifft_grad pair)
  // (fft 8192: the tail runs on 2 of the workgroup's 8 waves and measured +-0 / +2 % on the backward call, profiles/
  // r04_ab_launch_fusion.txt: kept on the separate kernel unless tuning flag 128 asks for it)
  if (dk && dk_done && a.nchunk == 1 && p->hp.N >= ((p->env_flags & 128) ? 8192 : 16384) && p->hp.N <= 32768 && p->hp.R == 1 &&
      !(p->env_flags & 32) && Lk > 0 && Lk <= p->hp.N) {
    d.dk_out = dk; d.Lk = (int)Lk; d.dk_scale = (float)(1.0 / p->hp.s_fwd);
    d.tab_bf = p->d_blob_bf; d.t_bf = p->hp_bf.tabs;      // (fp16 plans: bf16 tables for the tail)
    d.dk_fast = (Lk % 4 == 0) && !((uintptr_t)dk & 15);
    *dk_done = true;
  }
  // multi-pass sizes (fft 65536 / 131072, bf16 plans): every pass ends with its share of dk (Modes::dk_tail_rp); tuning flag 32 as above
  if (dk && dk_done && !*dk_done && a.nchunk == 1 && p->hp.R > 1 && p->hp.N1 > 1 && p->hp.dtype == DT_BF16 && !(p->env_flags & 32) &&
      Lk > 0 && Lk <= p->hp.N) {
    d.dk_out = dk; d.Lk = (int)Lk; d.dk_scale = (float)(1.0 / p->hp.s_fwd);
    d.tab_bf = p->d_blob_bf; d.t_bf = p->hp_bf.tabs;
    d.dk_fast = (Lk % 4 == 0) && !((uintptr_t)dk & 15);
    *dk_done = true;
  }
  // ... as complex rows (ffc_conv_bwd_kx: first step of dk at the HBM-level sizes; always bf16, the caller's scale)
  if (dk_pair && dk_done && a.nchunk == 1 && p->hp.N >= ((p->env_flags & 128) ? 8192 : 16384) && p->hp.N <= 32768 && p->hp.R == 1 &&
      !(p->env_flags & 32) && !((uintptr_t)dk_pair & 15)) {
    d.dk_pair = dk_pair; d.Lk = p->hp.N; d.dk_scale = dk_pair_scale; d.dk_fast = 1;
    d.tab_bf = p->d_blob_bf; d.t_bf = p->hp_bf.tabs;
    *dk_done = true;
  }
  d.dpost = (p->hp.N1 > 1 || yraw) ? dpost : nullptr;      // with y_raw every geometry writes dpost from its dout row load
  d.zin = zin;
  d.yraw = yraw;
  if (yraw && p->hp.R > 1 && p->hp.N1 > 1) {
    // multi-pass sizes of the fused 32768 kernel: the product runs as a streaming kernel of its own (see mul_rows_kernel)
    const int64_t HL = H * L;
    const int fast = (HL % 8 == 0) && !((sb_dout | sb_dpost) & 7) && !(((uintptr_t)dout | (uintptr_t)yraw | (uintptr_t)dpost) & 15);
    int64_t nb = (HL / (fast ? 8 : 1) + 255) / 256;
    if (nb > 8192) nb = 8192;
    const dim3 grid((unsigned)nb, (unsigned)B), block(256);
    if (p->hp.dtype == DT_BF16)
      hipLaunchKernelGGL(mul_rows_kernel<DT_BF16>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)dout, (const uint16_t*)yraw,
                         (uint16_t*)dpost, HL, sb_dout, HL, sb_dpost, fast);
    else
      hipLaunchKernelGGL(mul_rows_kernel<DT_F16>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)dout, (const uint16_t*)yraw,
                         (uint16_t*)dpost, HL, sb_dout, HL, sb_dpost, fast);
    HIPCHK(hipGetLastError());
    d.dpost = nullptr; d.yraw = nullptr; dpost = nullptr;
  }
  if (d.dpost && ((((uintptr_t)dpost) | (uintptr_t)yraw) & 15)) a.fast = 0;
#if defined(FFC_BWD_PROF)
  a.prof = ffc_bwd_prof_buffer();
#endif
  // fused sizes >= 4096 on saved spectra: the ZM = 1 kernels (ffc_k_bwdz.hip); everything else from this unit
  int rc = (zin && p->hp.N1 > 1) ? ffc_bwdz_launch(p->hp.N, p->hp.dtype, d, (hipStream_t)stream)
                                 : ffc_dispatch<BwdLaunchZ<0>::T>(p->hp.N, p->hp.dtype, d, (hipStream_t)stream);
  if (rc || !dpost || d.dpost) return rc;
  return ffc_conv_fwd_strided(p, u, kf, pregate, dout, dpost, B, H, L, 0, sb_u, sb_pre, sb_dout, sb_dpost, stream);
}
extern "C" int ffc_conv_bwd_gated_strided(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                                          const void* postgate, void* du, void* dpre, void* dpost, void* ws, int64_t B, int64_t H,
                                          int64_t L, int64_t sb_dout, int64_t sb_u, int64_t sb_pre, int64_t sb_post,
                                          int64_t sb_du, int64_t sb_dpre, int64_t sb_dpost, void* stream) {
  return conv_bwd_impl(p, dout, u, kf, pregate, postgate, du, dpre, dpost, ws, nullptr, B, H, L, sb_dout, sb_u, sb_pre, sb_post, sb_du,
                       sb_dpre, sb_dpost, stream);
}
// fused backward on the spectra saved by ffc_conv_fwd_z (same B, H, L, u, pregate): the first transform of every pair is skipped
extern "C" int ffc_conv_bwd_z(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                              const void* postgate, void* du, void* dpre, void* dpost, void* ws, const void* zin, int64_t B, int64_t H,
                              int64_t L, int64_t sb_dout, int64_t sb_u, int64_t sb_pre, int64_t sb_post,
                              int64_t sb_du, int64_t sb_dpre, int64_t sb_dpost, void* stream) {
  if (!zin) return ffc_fail("null spectrum buffer");
  return conv_bwd_impl(p, dout, u, kf, pregate, postgate, du, dpre, dpost, ws, zin, B, H, L, sb_dout, sb_u, sb_pre, sb_post, sb_du,
                       sb_dpre, sb_dpost, stream);
}
// ... and on the forward output before the postgate multiply that ffc_conv_fwd_z stored (y_raw, contiguous (B,H,L)): dpost =
// dout * y_raw is written from the registers that hold the dout rows (same fp32 product, rounded once, as the output gate)
extern "C" int ffc_conv_bwd_zy(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                               const void* postgate, void* du, void* dpre, void* dpost, void* ws, const void* zin, const void* y_raw,
                               int64_t B, int64_t H, int64_t L, int64_t sb_dout, int64_t sb_u, int64_t sb_pre, int64_t sb_post,
                               int64_t sb_du, int64_t sb_dpre, int64_t sb_dpost, void* stream) {
  if ((!zin && !(p && p->hp.N1 <= 1)) || !y_raw || !dpost) return ffc_fail("null spectrum / y_raw / dpost buffer (y_raw alone: single-tile sizes, fft <= 2048)");
  return conv_bwd_impl(p, dout, u, kf, pregate, postgate, du, dpre, dpost, ws, zin, B, H, L, sb_dout, sb_u, sb_pre, sb_post, sb_du,
                       sb_dpre, sb_dpost, stream, y_raw);
}
extern "C" int ffc_conv_bwd_gated(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                                  const void* postgate, void* du, void* dpre, void* dpost, void* ws, int64_t B, int64_t H,
                                  int64_t L, void* stream) {
  return ffc_conv_bwd_gated_strided(p, dout, u, kf, pregate, postgate, du, dpre, dpost, ws, B, H, L, 0, 0, 0, 0, 0, 0, 0, stream);
}

// The module's whole backward in one call: du (+ dpre, dpost) and dk (H, Lk) fp32; zin / y_raw: what ffc_conv_fwd_k saved (nullable:
// recomputing kernel).  dk comes out of the backward launch itself where a workgroup owns its head (Modes::dk_tail), through the
// fp32 slabs in ws + ffc_kernel_ifft_grad otherwise.
extern "C" int ffc_kernel_ifft_grad(const ffc_plan* p, const void* ws, int64_t B, int64_t H, int64_t Lk, float* dk, void* stream);
extern "C" int ffc_conv_bwd_k(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate, const void* postgate,
                              void* du, void* dpre, void* dpost, void* ws, const void* zin, const void* y_raw, float* dk, int64_t Lk,
                              int64_t B, int64_t H, int64_t L, void* stream) {
  if (!dk) return ffc_fail("null dk");
  bool dk_done = false;
  const void* yr = ((zin || p->hp.N1 <= 1) && y_raw && dpost) ? y_raw : nullptr;
  int rc = conv_bwd_impl(p, dout, u, kf, pregate, postgate, du, dpre, dpost, ws, zin, B, H, L, 0, 0, 0, 0, 0, 0, 0, stream, yr, dk, Lk,
                         &dk_done);
  if (rc || dk_done) return rc;
  return ffc_kernel_ifft_grad(p, ws, B, H, Lk, dk, stream);
}

// The backward of an HBM-level size's inner convolution in one call: input-gradient rows du and the dk rows as a complex pair-plane
// tensor (2, H, N) bf16 (as ffc_kernel_ifft_grad_c with `scale`), out of the backward launch itself where a workgroup owns its
// "head", through the fp32 slabs in ws otherwise.  zin: the spectra ffc_conv_fwd_kx kept (nullable).
extern "C" int ffc_kernel_ifft_grad_c(const ffc_plan* p, const void* ws, int64_t B, int64_t H, void* outpair, float scale, void* stream);
extern "C" int ffc_conv_bwd_kx(const ffc_plan* p, const void* dout, const void* u, const void* kf, void* du, void* ws, const void* zin,
                               void* outpair, float scale, int64_t B, int64_t H, int64_t L, void* stream) {
  if (!outpair) return ffc_fail("null outpair");
  bool dk_done = false;
  int rc = conv_bwd_impl(p, dout, u, kf, nullptr, nullptr, du, nullptr, nullptr, ws, zin, B, H, L, 0, 0, 0, 0, 0, 0, 0, stream, nullptr,
                         nullptr, 0, &dk_done, outpair, scale);
  if (rc || dk_done) return rc;
  return ffc_kernel_ifft_grad_c(p, ws, B, H, outpair, scale, stream);
}
