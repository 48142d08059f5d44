// dk_f accumulation kernel (Modes::dkf) + ffc_conv_bwd_dkf.
#ifndef FFC_GATE_BATCH
#define FFC_GATE_BATCH 1      // (see ffc_bwd_launch.h)
#endif
#include "ffc_dev.h"
using namespace ffc;

// Fused sizes keep the dk_f partial sums in a0..a127 (Modes::WAcc): the architectural half of the unified
// register file (128 VGPRs at 2 waves per SIMD) is what the allocator gets.
template <class GEO, int DT, bool HALF>
__global__ __launch_bounds__(GEO::WGW * 64, 2) __attribute__((amdgpu_num_vgpr(128))) void dkf_kernel(DkfArgs d) {
  if constexpr (GEO::NW == 1) {
    // one wave per unit (fft 4096): persistent workgroups walk the (head, chunk) jobs (see conv_kernel); the waves
    // only meet at the table copy and at the end-of-chunk reduction of the dk_f sums
    const int total = ((d.c.H + 7) & ~7) * d.c.nchunk;
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
      int h, chunk;
      if (map_id(id, d.c.H, d.c.nchunk, &h, &chunk)) Modes<DevBO, GEO, DT>::template dkf<HALF>(d, h, chunk, blockIdx.x);
    }
  } else {
    int h, chunk;
    if (!map_block(d.c.H, d.c.nchunk, &h, &chunk)) return;
    Modes<DevBO, GEO, DT>::template dkf<HALF>(d, h, chunk, blockIdx.x);
  }
}
template <class GEO, int DT>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void dkf_kernel_small(DkfArgs d) {   // single-tile sizes (N <= 1024)
  int h, chunk;
  if (!map_block(d.c.H, d.c.nchunk, &h, &chunk)) return;
  Modes<DevB, GEO, DT>::template dkf<false>(d, h, chunk, blockIdx.x);
}
// multi-pass sizes: the R passes of a (head, chunk) job run one after the other in the same workgroup (see conv_rp_kernel)
template <class GEO, int DT, bool HALF>
__global__ __launch_bounds__(GEO::WGW * 64, 2) __attribute__((amdgpu_num_vgpr(128))) void dkf_rp_kernel(DkfArgs d) {
  int h, chunk;
  if (!map_block(d.c.H, d.c.nchunk, &h, &chunk)) return;
  Modes<DevBO, GEO, DT>::BD::setup_tables(d.c.tab, d.c.t);
  const int wv = DevBO::wave(), wg = blockIdx.x;
#pragma unroll 1
  for (int k0 = 0; k0 < d.c.R; k0++) Modes<DevBO, GEO, DT>::template dkf<HALF, true>(d, h, chunk, wg, k0, wv);
}
template <class GEO, int DT>
struct DkfLaunch {
  static int run(const DkfArgs& d, hipStream_t st) {
    int hpad = (d.c.H + 7) & ~7;
    int ngrid = hpad * d.c.nchunk;
    if (d.c.R > 1) {
      if constexpr (GEO::N == 32768) {
        const dim3 grid(ngrid), block(GEO::WGW * 64);
        if (16 * GEO::Mi >= d.c.L) {
          int rc = ffc_set_lds(dkf_rp_kernel<GEO, DT, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((dkf_rp_kernel<GEO, DT, true>), grid, block, GEO::LDS_BYTES, st, d);
        } else {
          int rc = ffc_set_lds(dkf_rp_kernel<GEO, DT, false>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((dkf_rp_kernel<GEO, DT, false>), grid, block, GEO::LDS_BYTES, st, d);
        }
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("dkf_rp_kernel launch: ") + hipGetErrorString(e));
      } else if constexpr (GEO::OUTER) {
        return ffc_fail("multi-pass plan on a geometry without multi-pass kernels");
      }
    }
    if (GEO::OUTER && GEO::NW == 1 && d.c.persist > 0 && ngrid > d.c.persist) ngrid = d.c.persist;   // persistent: one per CU
    const dim3 grid(ngrid), block(GEO::WGW * 64);
    if constexpr (!GEO::OUTER) {
      using BD = Body<DevB, GEO, DT>;        // inner-only multi-pass form (fft 2048): per-pass tables behind the plan tables
      const int lds = GEO::LDS_BYTES + (d.c.R > 1 ? d.c.R * BD::IPASS_BYTES : 0);
      int rc = ffc_set_lds(dkf_kernel_small<GEO, DT>, GEO::LDS_BYTES + 2 * BD::IPASS_BYTES);
      if (rc) return rc;
      if (d.c.R > 1 && GEO::N != 1024) return ffc_fail("multi-pass plan on a geometry without multi-pass kernels");
      hipLaunchKernelGGL((dkf_kernel_small<GEO, DT>), grid, block, lds, st, d);
    } else {
      const bool half = (GEO::N1 / 2) * GEO::Mi >= d.c.L;
      if (half) {
        {
          int rc = ffc_set_lds(dkf_kernel<GEO, DT, true>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((dkf_kernel<GEO, DT, true>), grid, block, GEO::LDS_BYTES, st, d);
        }
      } else {
        int rc = ffc_set_lds(dkf_kernel<GEO, DT, false>, GEO::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL((dkf_kernel<GEO, DT, false>), grid, block, GEO::LDS_BYTES, st, d);
      }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffc_fail(std::string("dkf_kernel launch: ") + hipGetErrorString(e));
  }
};

extern "C" int64_t ffc_dkf_workspace_bytes(const ffc_plan* p, int64_t B, int64_t H) {
  if (!p) return 0;
  int nchunk, ppc;
  ffc_choose_chunks(p, (int)H, (int)((B + 1) / 2), &nchunk, &ppc);
  int upw = 8 / p->hp.NW;
  int64_t slabs = (int64_t)nchunk * ffc_slabs_per_chunk(p) * H * p->hp.R * p->hp.NT * 2048 * 4;   // rows (head, pass)
  // + spectrum scratch: one dtype-complex slot of the fused kernel's size per (workgroup, unit) of the grid
  int64_t hpad = (H + 7) & ~(int64_t)7;
  int64_t zs = p->hp.N1 > 1 ? hpad * nchunk * upw * (int64_t)(p->hp.N / p->hp.R) * 4 : 0;
  return slabs + zs;
}
// number of fp32 partial-sum slabs [slab][H][kf_elems][2] at the start of the workspace (k_f's internal order)
extern "C" int64_t ffc_dkf_slab_count(const ffc_plan* p, int64_t B, int64_t H) {
  if (!p) return 0;
  int nchunk, ppc;
  ffc_choose_chunks(p, (int)H, (int)((B + 1) / 2), &nchunk, &ppc);
  return (int64_t)nchunk * ffc_slabs_per_chunk(p);
}
extern "C" int ffc_conv_bwd_dkf(const ffc_plan* p, const void* dout, const void* u, const void* pregate, const void* postgate,
                                void* ws, int64_t B, int64_t H, int64_t L, void* stream) {
  if (!p || !dout || !u || !ws) return ffc_fail("null arg");
  if (B <= 0 || H <= 0) return ffc_fail("empty batch/heads");
  if (L <= 0 || L > p->hp.N) return ffc_fail("L must be in (0, fft_size]");
  if (B * H * L >= ((int64_t)1 << 31)) return ffc_fail("tensor too large (>= 2^31 elements)");
  DkfArgs d{};
  ConvArgs& a = d.c;
  a.u = u; a.pregate = pregate; a.postgate = postgate; a.tab = p->d_blob; a.t = p->hp.tabs;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.npair = (int)((B + 1) / 2); a.s_fwd = (float)p->hp.s_fwd;
  a.sbu = a.sbg = a.sbp = a.sby = H * L; d.sbd = d.sbdu = d.sbdpre = d.sbdpost = H * L;
  a.fast = (L % 8 == 0) && !(((uintptr_t)u | (uintptr_t)dout | (uintptr_t)pregate | (uintptr_t)postgate) & 15);
  ffc_choose_chunks(p, a.H, a.npair, &a.nchunk, &a.ppc);
  a.persist = ffc_persist(p);
  a.R = p->hp.R;
  a.stream = p->env_stream >= 0 ? p->env_stream : ((!pregate && !postgate && p->hp.R == 1) ? 1 : 0);    // see Body::STREAM_ROWS
  a.flags = p->env_flags;                        // tuning flags: 2 = k_f streamed, 4 = scratch streamed
  d.dout = dout; d.ws = (float*)ws; d.zscratch = ffc_zscratch(p, ws, a.H, a.nchunk);
  return ffc_dispatch<DkfLaunch>(p->hp.N, p->hp.dtype, d, (hipStream_t)stream);
}

