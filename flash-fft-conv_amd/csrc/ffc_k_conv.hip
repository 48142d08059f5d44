// FlashFFTConv forward / input-gradient kernel (see ffc_body.h) + ffc_conv_fwd.
#include "ffc_dev.h"
using namespace ffc;

// SZ: the forward that also stores the pairs' spectra for the backward pass (ConvArgs::zsave; fused sizes with an outer digit)
#ifndef FFC_SMALL_WAVES
#define FFC_SMALL_WAVES 2
#endif
template <class GEO, int DT, bool HALF, bool SZ = false, bool SP = false>
__global__ __launch_bounds__(GEO::WGW * 64, GEO::OUTER ? 2 : FFC_SMALL_WAVES) void conv_kernel(ConvArgs a) {
  using BD = Body<DevB, GEO, DT>;
  if constexpr (GEO::OUTER && GEO::NW == 1) {
    // One wave per unit (fft 4096): persistent workgroups, one per CU, walk the (head, chunk) jobs with a stride of
    // the grid (a multiple of 8, so a workgroup's heads stay on its XCD).  No phase of these units needs a
    // workgroup barrier (Body::unit_barrier), so the eight waves drift apart across jobs: one wave's row loads and
    // stores overlap another's transforms, and the plan tables are copied to LDS once per CU instead of once per head.
    BD::setup_tables(a.tab, a.t);
    const int total = ((a.H + 7) & ~7) * a.nchunk;
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
      int h, chunk;
      if (map_id(id, a.H, a.nchunk, &h, &chunk)) BD::template conv_job<HALF, false, SZ>(a, h, chunk);
    }
  } else if constexpr (!GEO::OUTER) {
    // single-tile sizes (fft <= 1024): persistent workgroups too (two per CU): a job is one tile per wave, copying the plan
    // tables to LDS for every job cost as much as the job
    BD::setup_tables(a.tab, a.t);
    const int total = ((a.H + 7) & ~7) * a.nchunk;
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
      int h, chunk;
      if (map_id(id, a.H, a.nchunk, &h, &chunk)) BD::template conv_job<HALF, false, SZ>(a, h, chunk);
    }
  } else {
    int h, chunk;
    if (!map_block(a.H, a.nchunk, &h, &chunk)) return;
    stagger_start(a.flags);
    if constexpr (GEO::NW > 1 && !SP) {
      // k -> k_f of this head first (ConvArgs::kfuse_k, one chunk per head): no separate launch for it
      BD::setup_tables(a.tab, a.t);
      if (a.kfuse_k || a.kfuse_x) Modes<DevB, GEO, DT>::kfft_head(a, h);
      BD::template conv_job<HALF, false, SZ, SP>(a, h, chunk);
    } else {
      BD::template conv<HALF, SZ, SP>(a, h, chunk);
    }
  }
}

// multi-pass sizes (fft 65536 / 131072): the R passes of a (head, chunk) job run one after the other in the same workgroup,
// so that the read-modify-write of the output rows stays inside one wave (struct Pass, Body::rows_out_rp)
template <class GEO, int DT, bool HALF, bool SZ = false>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void conv_rp_kernel(ConvArgs a) {
  using BD = Body<DevB, GEO, DT>;
  if constexpr (!GEO::OUTER) {
    // inner-only form (fft 2048): persistent workgroups, tables (incl. the per-pass ones) copied once
    BD::setup_tables(a.tab, a.t);
    BD::setup_tables_ipass(a.tab, a.t, a.R);
    const int total = ((a.H + 7) & ~7) * a.nchunk;
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
      int h, chunk;
      if (map_id(id, a.H, a.nchunk, &h, &chunk)) BD::template conv_job<HALF, true, SZ>(a, h, chunk);
    }
  } else {
    int h, chunk;
    if (!map_block(a.H, a.nchunk, &h, &chunk)) return;
    BD::setup_tables(a.tab, a.t);
    BD::template conv_job<HALF, true, SZ>(a, h, chunk);
  }
}

template <class GEO, int DT>
struct ConvLaunch {
  static int run(const ConvArgs& a, hipStream_t st) {
    using BD = Body<DevB, GEO, DT>;
    int hpad = (a.H + 7) & ~7;
    int grid = hpad * a.nchunk;
    if (a.R > 1) {
      if constexpr (GEO::N == 32768) {
        if (a.zsave) {
          if (16 * GEO::Mi >= a.L) {
            int rc = ffc_set_lds(conv_rp_kernel<GEO, DT, true, true>, GEO::LDS_BYTES);
            if (rc) return rc;
            hipLaunchKernelGGL((conv_rp_kernel<GEO, DT, true, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
          } else {
            int rc = ffc_set_lds(conv_rp_kernel<GEO, DT, false, true>, GEO::LDS_BYTES);
            if (rc) return rc;
            hipLaunchKernelGGL((conv_rp_kernel<GEO, DT, false, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
          }
          hipError_t e = hipGetLastError();
          return e == hipSuccess ? 0 : ffc_fail(std::string("conv_rp_kernel (spectrum-saving) launch: ") + hipGetErrorString(e));
        }
        if (16 * GEO::Mi >= a.L) {
          int rc = ffc_set_lds(conv_rp_kernel<GEO, DT, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_rp_kernel<GEO, DT, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
        } else {
          int rc = ffc_set_lds(conv_rp_kernel<GEO, DT, false>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_rp_kernel<GEO, DT, false>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
        }
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("conv_rp_kernel launch: ") + hipGetErrorString(e));
      } else if constexpr (GEO::N == 1024) {
        constexpr int lds = GEO::LDS_BYTES + 2 * BD::IPASS_BYTES;
        const int cap = (a.persist > 0 && a.persist < (1 << 29)) ? 2 * a.persist : (1 << 30);      // FFC_PERSIST=0: uncapped
        if (a.zsave || a.yraw) {
          int rc = ffc_set_lds(conv_rp_kernel<GEO, DT, false, true>, lds);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_rp_kernel<GEO, DT, false, true>), dim3(grid > cap ? cap : grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES + a.R * BD::IPASS_BYTES, st, a);
        } else {
          int rc = ffc_set_lds(conv_rp_kernel<GEO, DT, false>, lds);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_rp_kernel<GEO, DT, false>), dim3(grid > cap ? cap : grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES + a.R * BD::IPASS_BYTES, st, a);
        }
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("conv_rp_kernel launch: ") + hipGetErrorString(e));
      } else {
        return ffc_fail("multi-pass plan on a geometry without multi-pass kernels");
      }
    }
    if (GEO::OUTER && GEO::NW == 1 && grid > a.persist) grid = a.persist;      // persistent: one workgroup per CU
    if (!GEO::OUTER && a.persist > 0 && a.persist < (1 << 29) && grid > 2 * a.persist) grid = 2 * a.persist;      // persistent: two per CU
    if (a.sparse) {
      if constexpr (GEO::HAS_SP) {
        if ((GEO::N1 / 2) * GEO::Mi >= a.L) {
          int rc = ffc_set_lds(conv_kernel<GEO, DT, true, false, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_kernel<GEO, DT, true, false, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
        } else {
          int rc = ffc_set_lds(conv_kernel<GEO, DT, false, false, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_kernel<GEO, DT, false, false, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
        }
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("conv_kernel (frequency-sparse) launch: ") + hipGetErrorString(e));
      } else {
        return ffc_fail("frequency-sparse kernel: fft 16384 / 32768 only");
      }
    }
    if (a.zsave || (!GEO::OUTER && a.yraw)) {
      if constexpr (GEO::OUTER) {
        if ((GEO::N1 / 2) * GEO::Mi >= a.L) {
          int rc = ffc_set_lds(conv_kernel<GEO, DT, true, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_kernel<GEO, DT, true, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
        } else {
          int rc = ffc_set_lds(conv_kernel<GEO, DT, false, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((conv_kernel<GEO, DT, false, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
        }
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("conv_kernel (spectrum-saving) launch: ") + hipGetErrorString(e));
      } else {
        int rc = ffc_set_lds(conv_kernel<GEO, DT, false, true>, GEO::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL((conv_kernel<GEO, DT, false, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("conv_kernel (spectrum-saving, single tile) launch: ") + hipGetErrorString(e));
      }
    }
    // HALF variant (own register allocation): 32-point outer digit and L <= N/2, only E rows < 16 carry data
    if (GEO::OUTER && (GEO::N1 / 2) * GEO::Mi >= a.L) {
      int rc = ffc_set_lds(conv_kernel<GEO, DT, true>, GEO::LDS_BYTES);
      if (rc) return rc;
      hipLaunchKernelGGL((conv_kernel<GEO, DT, true>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
    } else {
      int rc = ffc_set_lds(conv_kernel<GEO, DT, false>, GEO::LDS_BYTES);
      if (rc) return rc;
      hipLaunchKernelGGL((conv_kernel<GEO, DT, false>), dim3(grid), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffc_fail(std::string("conv_kernel launch: ") + hipGetErrorString(e));
  }
};

// stride 0 = contiguous (H * L)
static inline bool ffc_stride_ok(int64_t* sb, int64_t B, int64_t H, int64_t L) {
  if (*sb == 0) *sb = H * L;
  return *sb >= H * L && (B - 1) * *sb + H * L < ((int64_t)1 << 31);
}

// spectra saved for the backward pass: [H][npair][N] complex values of the plan dtype (single-tile sizes: see below)
extern "C" int64_t ffc_spectrum_bytes(const ffc_plan* p, int64_t B, int64_t H) {
  if (!p || B <= 0 || H <= 0) return 0;
  if (p->hp.N1 <= 1) {      // single-tile sizes (fft <= 2048): [H][tiles of G pairs][R passes] slots of 1024 complex values
    const int64_t G = (32 / p->hp.N2) * (32 / p->hp.N3);
    return H * (((B + 1) / 2 + G - 1) / G) * p->hp.R * 4096;
  }
  return ((B + 1) / 2) * H * (int64_t)p->hp.N * 4;        // (hp.N = the plan's fft size, R passes x the kernel size)
}
static int conv_fwd_impl(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate,
                         void* y, void* zsave, void* yraw, int sparse, int64_t B, int64_t H, int64_t L, int conj_kf, int64_t sb_u, int64_t sb_pre,
                         int64_t sb_post, int64_t sb_y, void* stream, const float* kfuse_k = nullptr, int64_t kfuse_Lk = 0,
                         bool* kfuse_done = nullptr, const void* kfuse_x = nullptr, float kfuse_xscale = 1.0f) {
  if (!p || !u || !kf || !y) return ffc_fail("null arg");
  if (B <= 0 || H <= 0) return ffc_fail("empty batch/heads");
  if (L <= 0 || L > p->hp.N) return ffc_fail("L must be in (0, fft_size]");
  if ((uintptr_t)kf & 15) return ffc_fail("k_f must be 16-byte aligned");
  if (!ffc_stride_ok(&sb_u, B, H, L) || !ffc_stride_ok(&sb_pre, B, H, L) || !ffc_stride_ok(&sb_post, B, H, L) ||
      !ffc_stride_ok(&sb_y, B, H, L))
    return ffc_fail("tensor too large (>= 2^31 elements) or batch stride smaller than H*L");
  ConvArgs a{};
  a.u = u; a.pregate = pregate; a.postgate = postgate; a.y = y; a.kf = kf;
  a.tab = p->d_blob; a.t = p->hp.tabs;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.npair = (int)((B + 1) / 2);
  a.sbu = sb_u; a.sbg = sb_pre; a.sbp = sb_post; a.sby = sb_y;
  a.conj_kf = conj_kf;
  // y_raw without the spectra: the single-tile sizes (fft <= 2048) only -- their backward transforms u * pregate again from rows it
  // loads anyway (dpregate, du) and takes dpostgate = dout * y_raw from its dout row load (round 6)
  a.zsave = zsave; a.yraw = (zsave || p->hp.N1 <= 1) ? yraw : nullptr;
  a.sparse = sparse;
  if (sparse && (sparse < 0 || sparse > 4 || zsave || p->hp.R > 1 || p->hp.N2 != 32 || p->hp.N3 != 32 || p->hp.N1 <= 1))
    return ffc_fail("frequency-sparse forward: fft 16384 / 32768, 1 <= rows <= 4, no spectrum buffer");
  if (zsave && (ffc_spectrum_bytes(p, B, H) == 0 || ((uintptr_t)zsave & 15))) return ffc_fail("spectrum buffer: unsupported plan or misaligned");
  a.s_inv = (float)p->hp.s_inv; a.s_fwd = (float)p->hp.s_fwd;
  a.flags = p->env_flags;
  a.fast = (L % 8 == 0) && !(((uintptr_t)u | (uintptr_t)y | (uintptr_t)pregate | (uintptr_t)postgate) & 15) &&
           !((sb_u | sb_pre | sb_post | sb_y) & 7);
  ffc_choose_chunks(p, a.H, a.npair, &a.nchunk, &a.ppc, true);
  a.persist = ffc_persist(p);
  a.R = p->hp.R;
  // k -> k_f inside this launch (Modes::kfft_head): a workgroup owns its head, fft 8192 / 16384 / 32768 (the sizes whose waves
  // meet at workgroup barriers anyway); tuning flag 64 keeps the separate launch
  if (kfuse_k && kfuse_done && a.nchunk == 1 && p->hp.N >= 8192 && p->hp.N <= 32768 && p->hp.R == 1 && !sparse &&
      !(p->env_flags & 64) && kfuse_Lk > 0 && kfuse_Lk <= p->hp.N && H * kfuse_Lk < ((int64_t)1 << 31)) {
    a.kfuse_k = kfuse_k; a.kfuse_Lk = (int)kfuse_Lk;
    a.kfuse_scale = (float)(p->hp.s_k / p->hp.s_fwd) / (p->hp.dtype == DT_F16 ? 256.f : 1.f);
    a.kfuse_fast = (kfuse_Lk % 4 == 0) && !((uintptr_t)kfuse_k & 15);
    *kfuse_done = true;
  }
  // ... from complex rows (ffc_conv_fwd_kx: the inner k_f rows of the HBM-level sizes, the caller's scale)
  if (kfuse_x && kfuse_done && a.nchunk == 1 && p->hp.N >= 8192 && p->hp.N <= 32768 && p->hp.R == 1 && !sparse &&
      !(p->env_flags & 64) && !((uintptr_t)kfuse_x & 15)) {
    a.kfuse_x = kfuse_x; a.kfuse_scale = kfuse_xscale; a.kfuse_Lk = p->hp.N; a.kfuse_fast = 1;
    *kfuse_done = true;
  }
  // every row is read / written exactly once per launch (multi-pass sizes re-read the rows in every pass: plain accesses)
  a.stream = p->env_stream >= 0 ? p->env_stream : (p->hp.R > 1 ? 0 : 1);
  return ffc_dispatch<ConvLaunch>(p->hp.N, p->hp.dtype, a, (hipStream_t)stream);
}

extern "C" int ffc_conv_fwd_strided(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate,
                                    void* y, int64_t B, int64_t H, int64_t L, int conj_kf, int64_t sb_u, int64_t sb_pre,
                                    int64_t sb_post, int64_t sb_y, void* stream) {
  return conv_fwd_impl(p, u, kf, pregate, postgate, y, nullptr, nullptr, 0, B, H, L, conj_kf, sb_u, sb_pre, sb_post, sb_y, stream);
}
// forward that also stores every pair's spectrum FFT(u * pregate) in `zsave` (ffc_spectrum_bytes) for ffc_conv_bwd_z and,
// when y_raw is given (16-byte aligned, contiguous (B,H,L)), the output before the postgate multiply
extern "C" int ffc_conv_fwd_z(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate, void* y,
                              void* zsave, void* y_raw, int64_t B, int64_t H, int64_t L, int64_t sb_u, int64_t sb_pre, int64_t sb_post,
                              int64_t sb_y, void* stream) {
  if (!zsave && !(y_raw && p && p->hp.N1 <= 1)) return ffc_fail("null spectrum buffer (y_raw alone: single-tile sizes, fft <= 2048)");
  if (y_raw && ((uintptr_t)y_raw & 15)) return ffc_fail("y_raw must be 16-byte aligned");
  return conv_fwd_impl(p, u, kf, pregate, postgate, y, zsave, y_raw, 0, B, H, L, 0, sb_u, sb_pre, sb_post, sb_y, stream);
}

// Forward / input-gradient pass with a LOW-PASS k_f: every non-zero bin f has k3 = f / (N1 N2) < rows or >= 32 - rows
// (rows <= 4), i.e. |f| < rows * N / 32 (FrequencySparseFFTConv with N_partial <= N / 4).  Same result as ffc_conv_fwd on the
// same (masked) k_f; the kernel skips the all-zero spectrum rows: half of the k_f loads and of the k_f product, and one of
// the two K-steps of the first inverse stage (reference: the truncated kernels, monarch_cuda/monarch_fwd_complex.h:462-528).
extern "C" int ffc_conv_fwd_sparse(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate,
                                   void* y, int64_t B, int64_t H, int64_t L, int conj_kf, int rows, void* stream) {
  if (rows < 1) return ffc_fail("frequency-sparse forward: rows must be >= 1");
  return conv_fwd_impl(p, u, kf, pregate, postgate, y, nullptr, nullptr, rows, B, H, L, conj_kf, 0, 0, 0, 0, stream);
}

extern "C" int ffc_conv_fwd(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate, void* y,
                            int64_t B, int64_t H, int64_t L, int conj_kf, void* stream) {
  return ffc_conv_fwd_strided(p, u, kf, pregate, postgate, y, B, H, L, conj_kf, 0, 0, 0, 0, stream);
}

// ---- profiling build (N = 32768, bf16 only): same body with s_memtime phase counters
template <class GEO, int DT>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void conv_prof_kernel(ConvArgs a) {
  int h, chunk;
  if (!map_block(a.H, a.nchunk, &h, &chunk)) return;
  using BD = Body<DevB, GEO, DT>;
  BD::setup_tables(a.tab, a.t);
  const int wv = DevB::wave();
  typename BD::Unit un;
  un.wq = wv % GEO::NW;
  const int u = wv / GEO::NW;
  un.eb = u * GEO::EBYTES;
  const int p0 = chunk * a.ppc;
  int p1 = p0 + a.ppc;
  if (p1 > a.npair) p1 = a.npair;
  if (16 * GEO::Mi >= a.L) BD::template outer_jobs<true, true>(a, h, p0, p1, u, un, blockIdx.x);
  else BD::template outer_jobs<false, true>(a, h, p0, p1, u, un, blockIdx.x);
}

// prof: device buffer of gridDim*8*8 uint64 (zero-initialised by the caller); returns grid size in *grid_out
extern "C" int ffc_conv_fwd_prof(const ffc_plan* p, const void* u, const void* kf, void* y, int64_t B, int64_t H, int64_t L,
                                 unsigned long long* prof, int* grid_out, void* stream) {
  if (!p || p->hp.N != 32768 || p->hp.dtype != DT_BF16) return ffc_fail("prof build: N=32768 bf16 only");
  using GEO = Geo<32, 32, 32>;
  ConvArgs a{};
  a.u = u; a.y = y; a.kf = kf; a.tab = p->d_blob; a.t = p->hp.tabs;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.npair = (int)((B + 1) / 2); a.s_inv = (float)p->hp.s_inv; a.s_fwd = (float)p->hp.s_fwd;
  a.fast = (L % 8 == 0);
  a.sbu = a.sbg = a.sbp = a.sby = H * L;
  a.flags = p->env_flags;
  a.prof = prof;
  ffc_choose_chunks(p, a.H, a.npair, &a.nchunk, &a.ppc);
  int rc = ffc_set_lds(conv_prof_kernel<GEO, DT_BF16>, GEO::LDS_BYTES);
  if (rc) return rc;
  int hpad = (a.H + 7) & ~7;
  if (grid_out) *grid_out = hpad * a.nchunk;
  hipLaunchKernelGGL((conv_prof_kernel<GEO, DT_BF16>), dim3(hpad * a.nchunk), dim3(512), GEO::LDS_BYTES, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : ffc_fail(hipGetErrorString(e));
}

// The module's forward in one call (see ffc_hip.hip for the backward's): k (H, Lk) fp32 -> kf_out, then y = postgate * conv(u *
// pregate, k); zsave / y_raw as in ffc_conv_fwd_z (both nullable).  Where a workgroup owns its head the k -> k_f step runs inside
// the convolution launch (Modes::kfft_head); otherwise ffc_kernel_fft first.
extern "C" int ffc_kernel_fft(const ffc_plan* p, const float* k, int64_t H, int64_t Lk, void* kf, void* stream);
extern "C" int ffc_conv_fwd_k(const ffc_plan* p, const float* k, int64_t Lk, void* kf_out, const void* u, const void* pregate,
                              const void* postgate, void* y, void* zsave, void* y_raw, int64_t B, int64_t H, int64_t L, void* stream) {
  if (!p || !k || !kf_out) return ffc_fail("null arg");
  if (y_raw && ((uintptr_t)y_raw & 15)) return ffc_fail("y_raw must be 16-byte aligned");
  // decide first whether the convolution launch can take the k -> k_f step (same rule as conv_fwd_impl applies below)
  int nchunk = 0, ppc = 0;
  if (B > 0 && H > 0) ffc_choose_chunks(p, (int)H, (int)((B + 1) / 2), &nchunk, &ppc, true);
  const bool fuse = nchunk == 1 && p->hp.N >= 8192 && p->hp.N <= 32768 && p->hp.R == 1 && !(p->env_flags & 64) && Lk > 0 &&
                    Lk <= p->hp.N && H * Lk < ((int64_t)1 << 31);
  if (!fuse) {
    int rc = ffc_kernel_fft(p, k, H, Lk, kf_out, stream);
    if (rc) return rc;
  }
  bool done = false;
  int rc = conv_fwd_impl(p, u, kf_out, pregate, postgate, y, zsave, y_raw, 0, B, H, L, 0, 0, 0, 0, 0, stream,
                         fuse ? k : nullptr, Lk, &done);
  if (rc) return rc;
  if (fuse && !done) return ffc_fail("internal: the convolution launch did not take the k -> k_f step");
  return 0;
}

// The forward of an HBM-level size's inner convolution in one call: the inner k_f rows from their complex input (pair-plane tensor
// (2, H, N), as ffc_kernel_fft_c; `scale` = that call's scale) into kf_out, then the convolution of the rows `u` (no gates at this
// level), spectra kept in zsave when given.  k -> k_f runs inside the convolution launch where a workgroup owns its "head".
extern "C" int ffc_kernel_fft_c(const ffc_plan* p, const void* xpair, int64_t H, void* kf, float scale, void* stream);
extern "C" int ffc_conv_fwd_kx(const ffc_plan* p, const void* xpair, float scale, void* kf_out, const void* u, void* y, void* zsave,
                               int64_t B, int64_t H, int64_t L, void* stream) {
  if (!p || !xpair || !kf_out) return ffc_fail("null arg");
  int nchunk = 0, ppc = 0;
  if (B > 0 && H > 0) ffc_choose_chunks(p, (int)H, (int)((B + 1) / 2), &nchunk, &ppc, true);
  const bool fuse = nchunk == 1 && p->hp.N >= 8192 && p->hp.N <= 32768 && p->hp.R == 1 && !(p->env_flags & 64) && !((uintptr_t)xpair & 15);
  if (!fuse) {
    int rc = ffc_kernel_fft_c(p, xpair, H, kf_out, scale, stream);
    if (rc) return rc;
  }
  bool done = false;
  int rc = conv_fwd_impl(p, u, kf_out, nullptr, nullptr, y, zsave, nullptr, 0, B, H, L, 0, 0, 0, 0, 0, stream, nullptr, 0, &done,
                         fuse ? xpair : nullptr, scale);
  if (rc) return rc;
  if (fuse && !done) return ffc_fail("internal: the convolution launch did not take the k -> k_f step");
  return 0;
}
