// Fused backward kernel (Modes::bwd) + ffc_conv_bwd / ffc_conv_bwd_gated(_strided).  (Split from ffc_k_dkf.hip so that the two
// longest translation units of the library compile side by side.)
#include "ffc_dev.h"
using namespace ffc;

template <class GEO, int DT, bool HALF>
__global__ __launch_bounds__(GEO::WGW * 64, 2) __attribute__((amdgpu_num_vgpr(128))) void bwd_kernel(DkfArgs d) {
  if constexpr (GEO::NW == 1) {
    // one wave per unit (fft 4096): persistent workgroups walk the (head, chunk) jobs (see conv_kernel); the waves
    // only meet at the table copy and at the end-of-chunk reduction of the dk_f sums
    const int total = ((d.c.H + 7) & ~7) * d.c.nchunk;
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
      int h, chunk;
      if (map_id(id, d.c.H, d.c.nchunk, &h, &chunk)) Modes<DevBO, GEO, DT>::template bwd<HALF>(d, h, chunk, blockIdx.x);
    }
  } else {
    int h, chunk;
    if (!map_block(d.c.H, d.c.nchunk, &h, &chunk)) return;
    stagger_start(d.c.flags);
    Modes<DevBO, GEO, DT>::template bwd<HALF>(d, h, chunk, blockIdx.x);
  }
}
// single-tile sizes (fft <= 2048): persistent workgroups (two per CU) walk the (head, chunk) jobs, the plan tables are copied
// to LDS once per workgroup instead of once per job (a job is one pair per wave at B = 16: the copy was as large as the work)
template <class GEO, int DT>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void bwd_kernel_small(DkfArgs d) {
  using M = Modes<DevB, GEO, DT>;
  M::BD::setup_tables(d.c.tab, d.c.t);
  if constexpr (GEO::N == 1024) { if (d.c.R > 1) M::BD::setup_tables_ipass(d.c.tab, d.c.t, d.c.R); }
  const int total = ((d.c.H + 7) & ~7) * d.c.nchunk;
  for (int id = blockIdx.x; id < total; id += gridDim.x) {
    int h, chunk;
    if (map_id(id, d.c.H, d.c.nchunk, &h, &chunk)) M::template bwd<false, false, false>(d, h, chunk, id);
  }
}
template <class GEO, int DT, bool HALF>
__global__ __launch_bounds__(GEO::WGW * 64, 2) __attribute__((amdgpu_num_vgpr(128))) void bwd_rp_kernel(DkfArgs d) {
  int h, chunk;
  if (!map_block(d.c.H, d.c.nchunk, &h, &chunk)) return;
  Modes<DevBO, GEO, DT>::BD::setup_tables(d.c.tab, d.c.t);
  const int wv = DevBO::wave(), wg = blockIdx.x;
#pragma unroll 1
  for (int k0 = 0; k0 < d.c.R; k0++) Modes<DevBO, GEO, DT>::template bwd<HALF, true>(d, h, chunk, wg, k0, wv);
}

template <class GEO, int DT>
struct BwdLaunch {
  static int run(const DkfArgs& d, hipStream_t st) {
    int hpad = (d.c.H + 7) & ~7;
    int ngrid = hpad * d.c.nchunk;
    if (d.c.R > 1) {
      if constexpr (GEO::N == 32768) {
        const dim3 grid(ngrid), block(GEO::WGW * 64);
        if (16 * GEO::Mi >= d.c.L) {
          int rc = ffc_set_lds(bwd_rp_kernel<GEO, DT, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((bwd_rp_kernel<GEO, DT, true>), grid, block, GEO::LDS_BYTES, st, d);
        } else {
          int rc = ffc_set_lds(bwd_rp_kernel<GEO, DT, false>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((bwd_rp_kernel<GEO, DT, false>), grid, block, GEO::LDS_BYTES, st, d);
        }
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("bwd_rp_kernel launch: ") + hipGetErrorString(e));
      } else if constexpr (GEO::OUTER) {
        return ffc_fail("multi-pass plan on a geometry without multi-pass kernels");
      }
    }
    if (GEO::OUTER && GEO::NW == 1 && d.c.persist > 0 && ngrid > d.c.persist) ngrid = d.c.persist;   // persistent: one per CU
    const dim3 grid(ngrid), block(GEO::WGW * 64);
    if constexpr (!GEO::OUTER) {
      using BD = Body<DevB, GEO, DT>;
      const int lds = GEO::LDS_BYTES + (d.c.R > 1 ? d.c.R * BD::IPASS_BYTES : 0);
      int rc = ffc_set_lds(bwd_kernel_small<GEO, DT>, GEO::LDS_BYTES + 2 * BD::IPASS_BYTES);
      if (rc) return rc;
      if (d.c.R > 1 && GEO::N != 1024) return ffc_fail("multi-pass plan on a geometry without multi-pass kernels");
      const int cap = (d.c.persist > 0 && d.c.persist < (1 << 29)) ? 2 * d.c.persist : (1 << 30);      // FFC_PERSIST=0: uncapped
      hipLaunchKernelGGL((bwd_kernel_small<GEO, DT>), dim3(ngrid > cap ? cap : ngrid), block, lds, st, d);
    } else {
      const bool half = (GEO::N1 / 2) * GEO::Mi >= d.c.L;
      if (half) {
        {
          int rc = ffc_set_lds(bwd_kernel<GEO, DT, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((bwd_kernel<GEO, DT, true>), grid, block, GEO::LDS_BYTES, st, d);
        }
      } else {
        int rc = ffc_set_lds(bwd_kernel<GEO, DT, false>, GEO::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL((bwd_kernel<GEO, DT, false>), grid, block, GEO::LDS_BYTES, st, d);
      }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffc_fail(std::string("bwd_kernel launch: ") + hipGetErrorString(e));
  }
};

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
                         int64_t sb_du, int64_t sb_dpre, int64_t sb_dpost, void* stream) {
  if (!p || !dout || !u || !kf || !du || !ws) return ffc_fail("null arg");
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
  d.dpost = p->hp.N1 > 1 ? dpost : nullptr;
  d.zin = zin;
  if (d.dpost && (((uintptr_t)dpost) & 15)) a.fast = 0;
#if defined(FFC_BWD_PROF)
  a.prof = ffc_bwd_prof_buffer();
#endif
  int rc = ffc_dispatch<BwdLaunch>(p->hp.N, p->hp.dtype, d, (hipStream_t)stream);
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
extern "C" int ffc_conv_bwd_gated(const ffc_plan* p, const void* dout, const void* u, const void* kf, const void* pregate,
                                  const void* postgate, void* du, void* dpre, void* dpost, void* ws, int64_t B, int64_t H,
                                  int64_t L, void* stream) {
  return ffc_conv_bwd_gated_strided(p, dout, u, kf, pregate, postgate, du, dpre, dpost, ws, B, H, L, 0, 0, 0, 0, 0, 0, 0, stream);
}
