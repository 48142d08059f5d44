// dk_f -> dk kernel (Modes::dkifft) + ffc_kernel_ifft_grad.
#include "ffc_dev.h"
using namespace ffc;

template <class GEO, int DT>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void dkifft_kernel(DkArgs a) {
  Modes<DevB, GEO, DT>::dkifft(a, blockIdx.x);
}
template <class GEO, int DT>
struct DkLaunch {
  static int run(const DkArgs& a, hipStream_t st) {
    using BD = Body<DevB, GEO, DT>;
    const int lds = GEO::LDS_BYTES + ((!GEO::OUTER && a.R > 1) ? a.R * BD::IPASS_BYTES : 0);   // inner-only multi-pass tables
    int rc = ffc_set_lds(dkifft_kernel<GEO, DT>, GEO::LDS_BYTES + (GEO::OUTER ? 0 : 2 * BD::IPASS_BYTES));
    if (rc) return rc;
    const int nunits = GEO::OUTER ? a.H : (a.H + GEO::G - 1) / GEO::G;
    hipLaunchKernelGGL((dkifft_kernel<GEO, DT>), dim3((nunits + GEO::UPW - 1) / GEO::UPW), dim3(GEO::WGW * 64), lds, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffc_fail(std::string("dkifft_kernel launch: ") + hipGetErrorString(e));
  }
};

// dk from `nslab` explicit fp32 slabs [nslab][H][kf_elems][2] (k_f's internal order): the entry point of callers that
// reduce the partial sums themselves (B-shard multi-GPU: reduce-scatter over ranks, then one slab of H/W heads)
extern "C" int ffc_kernel_ifft_grad_slabs(const ffc_plan* p, const void* slabs, int64_t nslab, int64_t H, int64_t Lk, float* dk,
                                          void* stream) {
  if (!p || !slabs || !dk) return ffc_fail("null arg");
  if (H <= 0 || Lk <= 0 || Lk > p->hp.N) return ffc_fail("dk must be (H, Lk) with 0 < Lk <= fft_size");
  if (nslab <= 0 || nslab > 65536) return ffc_fail("bad slab count");
  if ((uintptr_t)slabs & 15) return ffc_fail("slabs must be 16-byte aligned");
  DkArgs a{};
  // always bf16 arithmetic: W/N underflows fp16 for small gradients, bf16 keeps fp32's range
  a.ws = (const float*)slabs; a.dk = dk; a.tab = p->d_blob_bf; a.t = p->hp_bf.tabs; a.H = (int)H; a.Lk = (int)Lk;
  a.nslab = (int)nslab;
  a.scale = (float)(1.0 / p->hp.s_fwd); a.s_inv = (float)p->hp_bf.s_inv;   // tile_inv applies s_inv = 1/(N s_fwd)
  a.fast = (Lk % 4 == 0) && !((uintptr_t)dk & 15);
  a.flags = p->env_flags;
  a.R = p->hp.R;
  return ffc_dispatch<DkLaunch>(p->hp.N, DT_BF16, a, (hipStream_t)stream);
}

extern "C" int ffc_kernel_ifft_grad(const ffc_plan* p, const void* ws, int64_t B, int64_t H, int64_t Lk, float* dk, void* stream) {
  if (!p || !ws || !dk) return ffc_fail("null arg");
  if (H <= 0 || Lk <= 0 || Lk > p->hp.N) return ffc_fail("dk must be (H, Lk) with 0 < Lk <= fft_size");
  int nchunk, ppc;
  ffc_choose_chunks(p, (int)H, (int)((B + 1) / 2), &nchunk, &ppc);
  DkArgs a{};
  // always bf16 arithmetic: W/N underflows fp16 for small gradients, bf16 keeps fp32's range
  a.ws = (const float*)ws; a.dk = dk; a.tab = p->d_blob_bf; a.t = p->hp_bf.tabs; a.H = (int)H; a.Lk = (int)Lk;
  a.nslab = nchunk * ffc_slabs_per_chunk(p);
  a.scale = (float)(1.0 / p->hp.s_fwd); a.s_inv = (float)p->hp_bf.s_inv;   // tile_inv applies s_inv = 1/(N s_fwd)
  a.fast = (Lk % 4 == 0) && !((uintptr_t)dk & 15);
  a.flags = p->env_flags;
  a.R = p->hp.R;
  return ffc_dispatch<DkLaunch>(p->hp.N, DT_BF16, a, (hipStream_t)stream);
}

// complex output (pair-plane tensor (2, H, N) bf16) instead of dk: first step of dk for big FFT sizes
extern "C" int ffc_kernel_ifft_grad_c(const ffc_plan* p, const void* ws, int64_t B, int64_t H, void* outpair, float scale, void* stream) {
  if (!p || !ws || !outpair) return ffc_fail("null arg");
  if (p->hp.N1 <= 1) return ffc_fail("ffc_kernel_ifft_grad_c: inner size must be >= 4096");
  int nchunk, ppc;
  ffc_choose_chunks(p, (int)H, (int)((B + 1) / 2), &nchunk, &ppc);
  DkArgs a{};
  a.ws = (const float*)ws; a.outpair = outpair; a.tab = p->d_blob_bf; a.t = p->hp_bf.tabs; a.H = (int)H; a.Lk = p->hp.N;
  a.nslab = nchunk * ffc_slabs_per_chunk(p); a.scale = scale; a.s_inv = (float)p->hp_bf.s_inv; a.fast = 1;
  a.R = p->hp.R;
  return ffc_dispatch<DkLaunch>(p->hp.N, DT_BF16, a, (hipStream_t)stream);
}

// the same from `nslab` caller-owned fp32 slabs [nslab][H][kf_elems][2] (B-shard of the big FFT sizes: rows reduce-scattered
// over the ranks, flashfftconv/sharding.py)
extern "C" int ffc_kernel_ifft_grad_c_slabs(const ffc_plan* p, const void* slabs, int64_t nslab, int64_t H, void* outpair, float scale,
                                            void* stream) {
  if (!p || !slabs || !outpair) return ffc_fail("null arg");
  if (p->hp.N1 <= 1) return ffc_fail("ffc_kernel_ifft_grad_c_slabs: inner size must be >= 4096");
  if (H <= 0 || nslab <= 0 || nslab > 65536) return ffc_fail("bad head / slab count");
  if ((uintptr_t)slabs & 15) return ffc_fail("slabs must be 16-byte aligned");
  DkArgs a{};
  a.ws = (const float*)slabs; a.outpair = outpair; a.tab = p->d_blob_bf; a.t = p->hp_bf.tabs; a.H = (int)H; a.Lk = p->hp.N;
  a.nslab = (int)nslab; a.scale = scale; a.s_inv = (float)p->hp_bf.s_inv; a.fast = 1;
  a.R = p->hp.R;
  return ffc_dispatch<DkLaunch>(p->hp.N, DT_BF16, a, (hipStream_t)stream);
}
