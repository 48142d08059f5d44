// Fused backward kernels (Modes::bwd) and their launcher, shared by the two translation units that instantiate them:
//   ffc_k_bwd.hip   ZM = 0: recomputing form (+ the single-tile kernels, which decide at run time)
//   ffc_k_bwdz.hip  ZM = 1: on the spectra saved by the forward pass (ffc_conv_bwd_z / ffc_conv_bwd_zy)
// Two units so that the library's longest compile runs as two parallel halves, and so that a profile names the two forms apart
// (one kernel name averaged saved-spectra and recompute launches).
#pragma once
// gated rows: gate loads one chunk (both planes: 2 x 16 bytes per lane in flight) at a time: these kernels run on a 128-VGPR budget next
// to the 128 accumulation registers; with batches of 2 chunks the allocator reached into a0..a127 (build.py check_agpr), the forward
// kernels take 4 (ffc_body.h rows_store_g)
#ifndef FFC_GATE_BATCH
#define FFC_GATE_BATCH 1
#define FFC_GATE_BATCH_OUT 2
#endif
#include "ffc_dev.h"
using namespace ffc;

template <class GEO, int DT, bool HALF, int ZM>
__global__ __launch_bounds__(GEO::WGW * 64, 2) __attribute__((amdgpu_num_vgpr(128))) void bwd_kernel(DkfArgs d) {
  if constexpr (GEO::NW == 1) {
    // one wave per unit (fft 4096): persistent workgroups walk the (head, chunk) jobs (see conv_kernel); the waves
    // only meet at the table copy and at the end-of-chunk reduction of the dk_f sums
    const int total = ((d.c.H + 7) & ~7) * d.c.nchunk;
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
      int h, chunk;
      if (map_id(id, d.c.H, d.c.nchunk, &h, &chunk)) Modes<DevBO, GEO, DT>::template bwd<HALF, false, true, ZM>(d, h, chunk, blockIdx.x);
    }
  } else {
    int h, chunk;
    if (!map_block(d.c.H, d.c.nchunk, &h, &chunk)) return;
    stagger_start(d.c.flags);
    Modes<DevBO, GEO, DT>::template bwd<HALF, false, true, ZM>(d, h, chunk, blockIdx.x);
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
// FASTK: launched only when every tensor is 16-byte aligned and L % 8 == 0 (DkfArgs::c.fast): the row code exists once, as 16-byte
// accesses (DevBOF); FASTK = false keeps the run-time switch for ragged / misaligned calls
template <class GEO, int DT, bool HALF, int ZM, bool FASTK>
__global__ __launch_bounds__(GEO::WGW * 64, 2) __attribute__((amdgpu_num_vgpr(128))) void bwd_rp_kernel(DkfArgs d) {
  using BK = typename std::conditional<FASTK, DevBOF, DevBO>::type;
  int h, chunk;
  if (!map_block(d.c.H, d.c.nchunk, &h, &chunk)) return;
  Modes<BK, GEO, DT>::BD::setup_tables(d.c.tab, d.c.t);
  const int wv = DevBO::wave(), wg = blockIdx.x;
#pragma unroll 1
  for (int k0 = 0; k0 < d.c.R; k0++) Modes<BK, GEO, DT>::template bwd<HALF, true, true, ZM>(d, h, chunk, wg, k0, wv);
}
template <class K>
static int ffc_bwd_rp_go(K kernel, int lds, dim3 grid, dim3 block, hipStream_t st, const DkfArgs& d) {
  int rc = ffc_set_lds(kernel, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(kernel, grid, block, lds, st, d);
  return 0;
}

template <int ZM>
struct BwdLaunchZ {
template <class GEO, int DT>
struct T {
  static int run(const DkfArgs& d, hipStream_t st) {
    int hpad = (d.c.H + 7) & ~7;
    int ngrid = hpad * d.c.nchunk;
    if (d.c.R > 1) {
      if constexpr (GEO::N == 32768) {
        const dim3 grid(ngrid), block(GEO::WGW * 64);
        const bool half = 16 * GEO::Mi >= d.c.L;
        int rc;
        if (d.c.fast && (FFC_RP_FASTK != 0)) rc = half ? ffc_bwd_rp_go(bwd_rp_kernel<GEO, DT, true, ZM, true>, GEO::LDS_BYTES, grid, block, st, d)
                                : ffc_bwd_rp_go(bwd_rp_kernel<GEO, DT, false, ZM, true>, GEO::LDS_BYTES, grid, block, st, d);
        else rc = half ? ffc_bwd_rp_go(bwd_rp_kernel<GEO, DT, true, ZM, false>, GEO::LDS_BYTES, grid, block, st, d)
                       : ffc_bwd_rp_go(bwd_rp_kernel<GEO, DT, false, ZM, false>, GEO::LDS_BYTES, grid, block, st, d);
        if (rc) return rc;
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : ffc_fail(std::string("bwd_rp_kernel launch: ") + hipGetErrorString(e));
      } else if constexpr (GEO::OUTER) {
        return ffc_fail("multi-pass plan on a geometry without multi-pass kernels");
      }
    }
    if (GEO::OUTER && GEO::NW == 1 && d.c.persist > 0 && ngrid > d.c.persist) ngrid = d.c.persist;   // persistent: one per CU
    const dim3 grid(ngrid), block(GEO::WGW * 64);
    if constexpr (!GEO::OUTER && ZM != 0) {
      return ffc_fail("single-tile sizes are launched by the recomputing unit (run-time spectrum switch)");
    } else if constexpr (!GEO::OUTER) {
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
          int rc = ffc_set_lds(bwd_kernel<GEO, DT, true, ZM>, GEO::LDS_BYTES);
          if (rc) return rc;
          hipLaunchKernelGGL((bwd_kernel<GEO, DT, true, ZM>), grid, block, GEO::LDS_BYTES, st, d);
        }
      } else {
        int rc = ffc_set_lds(bwd_kernel<GEO, DT, false, ZM>, GEO::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL((bwd_kernel<GEO, DT, false, ZM>), grid, block, GEO::LDS_BYTES, st, d);
      }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffc_fail(std::string("bwd_kernel launch: ") + hipGetErrorString(e));
  }
};
};

