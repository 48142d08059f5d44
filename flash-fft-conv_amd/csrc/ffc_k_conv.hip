// FlashFFTConv forward / input-gradient kernel (see ffc_body.h) + ffc_conv_fwd.
#include "ffc_dev.h"
using namespace ffc;

template <class GEO, int DT>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void conv_kernel(ConvArgs a) {
  int h, chunk;
  if (!map_block(a.H, a.nchunk, &h, &chunk)) return;
  Body<DevB, GEO, DT>::conv(a, h, chunk);
}

template <class GEO, int DT>
struct ConvLaunch {
  static int run(const ConvArgs& a, hipStream_t st) {
    static int rc = ffc_set_lds(conv_kernel<GEO, DT>, GEO::LDS_BYTES);
    if (rc) return rc;
    int hpad = (a.H + 7) & ~7;
    hipLaunchKernelGGL((conv_kernel<GEO, DT>), dim3(hpad * a.nchunk), dim3(GEO::WGW * 64), GEO::LDS_BYTES, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffc_fail(std::string("conv_kernel launch: ") + hipGetErrorString(e));
  }
};

extern "C" int ffc_conv_fwd(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate, void* y,
                            int64_t B, int64_t H, int64_t L, int conj_kf, void* stream) {
  if (!p || !u || !kf || !y) return ffc_fail("null arg");
  if (B <= 0 || H <= 0) return ffc_fail("empty batch/heads");
  if (L <= 0 || L > p->hp.N) return ffc_fail("L must be in (0, fft_size]");
  if ((uintptr_t)kf & 15) return ffc_fail("k_f must be 16-byte aligned");
  if (B * H * L >= ((int64_t)1 << 31)) return ffc_fail("tensor too large (>= 2^31 elements)");
  ConvArgs a{};
  a.u = u; a.pregate = pregate; a.postgate = postgate; a.y = y; a.kf = kf;
  a.tab = p->d_blob; a.t = p->hp.tabs;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.npair = (int)((B + 1) / 2);
  a.conj_kf = conj_kf;
  a.s_inv = (float)p->hp.s_inv;
  a.fast = (L % 8 == 0) && !(((uintptr_t)u | (uintptr_t)y | (uintptr_t)pregate | (uintptr_t)postgate) & 15);
  ffc_choose_chunks(p, a.H, a.npair, &a.nchunk, &a.ppc);
  return ffc_dispatch<ConvLaunch>(p->hp.N, p->hp.dtype, a, (hipStream_t)stream);
}
