// k -> k_f kernel (Modes::kfft) + ffc_kernel_fft.
#include "ffc_dev.h"
using namespace ffc;

template <class GEO, int DT>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void kfft_kernel(KfArgs a) {
  Modes<DevB, GEO, DT>::kfft(a, blockIdx.x);
}
template <class GEO, int DT>
struct KfLaunch {
  static int run(const KfArgs& a, hipStream_t st) {
    using BD = Body<DevB, GEO, DT>;
    const int lds = GEO::LDS_BYTES + ((!GEO::OUTER && a.R > 1) ? a.R * BD::IPASS_BYTES : 0);   // inner-only multi-pass tables
    int rc = ffc_set_lds(kfft_kernel<GEO, DT>, GEO::LDS_BYTES + (GEO::OUTER ? 0 : 2 * BD::IPASS_BYTES));
    if (rc) return rc;
    const int nunits = GEO::OUTER ? a.H : (a.H + GEO::G - 1) / GEO::G;
    hipLaunchKernelGGL((kfft_kernel<GEO, DT>), dim3((nunits + GEO::UPW - 1) / GEO::UPW), dim3(GEO::WGW * 64), lds, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffc_fail(std::string("kfft_kernel launch: ") + hipGetErrorString(e));
  }
};

extern "C" int ffc_kernel_fft(const ffc_plan* p, const float* k, int64_t H, int64_t Lk, void* kf, void* stream) {
  if (!p || !k || !kf) return ffc_fail("null arg");
  if (H <= 0 || Lk <= 0 || Lk > p->hp.N) return ffc_fail("k must be (H, Lk) with 0 < Lk <= fft_size");
  if (H * Lk >= ((int64_t)1 << 31)) return ffc_fail("k too large");
  KfArgs a{};
  a.k = k; a.kf = kf; a.tab = p->d_blob; a.t = p->hp.tabs; a.H = (int)H; a.Lk = (int)Lk;
  a.s_fwd = (float)p->hp.s_fwd;
  a.prescale = p->hp.dtype == DT_F16 ? 256.f : 1.f;
  a.scale = (float)(p->hp.s_k / p->hp.s_fwd) / a.prescale;
  a.fast = (Lk % 4 == 0) && !((uintptr_t)k & 15);
  a.R = p->hp.R;
  return ffc_dispatch<KfLaunch>(p->hp.N, p->hp.dtype, a, (hipStream_t)stream);
}

// complex input (pair-plane tensor (2, H, N) dtype) instead of real k: inner k_f rows of big FFT sizes
extern "C" int ffc_kernel_fft_c(const ffc_plan* p, const void* xpair, int64_t H, void* kf, float scale, void* stream) {
  if (!p || !xpair || !kf) return ffc_fail("null arg");
  if (p->hp.N1 <= 1) return ffc_fail("ffc_kernel_fft_c: inner size must be >= 4096");
  KfArgs a{};
  a.xpair = xpair; a.kf = kf; a.tab = p->d_blob; a.t = p->hp.tabs; a.H = (int)H; a.Lk = p->hp.N; a.scale = scale; a.prescale = 1.f; a.s_fwd = (float)p->hp.s_fwd; a.fast = 1;
  a.R = p->hp.R;
  return ffc_dispatch<KfLaunch>(p->hp.N, p->hp.dtype, a, (hipStream_t)stream);
}
