// Phase-B occupancy probe (fft 32768, bf16): the REAL inner-tile code of csrc/ffc_body.h in a loop on LDS-resident data,
// in the register / occupancy regimes a kernel could run it in:
//   A  8 waves x 256 registers, two tiles in lock-step per wave (inner_tile2: the shipped forward kernel)
//   B 16 waves x 128 registers, one tile at a time per wave, inner twiddle streamed from LDS
//   C  8 waves x 128 registers, one tile at a time (the fused backward's regime, without its dk_f accumulation)
//   D  8 waves x 256 registers, one tile at a time, inner twiddle resident
// Every CU runs one workgroup (LDS: the 128 KB exchange buffer); each wave loops over its tiles `iters` times.
// Output: shader cycles per tile per CU-SIMD and ns per tile per CU.  Timing experiment only (k_f = unit-modulus noise).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm --amdgpu-mfma-vgpr-form phaseb_probe.hip \
//        ../../flash-fft-conv_amd/csrc/ffc_plan.cpp -o phaseb_probe
#include "../../flash-fft-conv_amd/csrc/ffc_dev.h"

#include <algorithm>
#include <vector>
using namespace ffc;

void ffc_set_error_(const char* m) { printf("error: %s\n", m); }

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

using GEO = Geo<32, 32, 32>;

template <class B>
__device__ __forceinline__ void fill_e(int nthreads) {
  // pseudo-random bf16 values of magnitude ~1 in both planes
  for (int i = threadIdx.x; i < GEO::EBYTES / 4; i += nthreads) {
    uint32_t x = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
    uint32_t lo = 0x3f00u | ((x >> 3) & 0x80ffu), hi = 0x3f00u | ((x >> 13) & 0x80ffu);
    B::lds_w32(i * 4, lo | (hi << 16));
  }
}

template <int MODE>
__global__ __launch_bounds__(MODE == 1 ? 1024 : 512, MODE == 1 ? 4 : (MODE == 2 ? 4 : 2)) void pb_kernel(ConvArgs a, int iters, unsigned long long* cyc) {
  using BK = typename std::conditional<MODE == 1 || MODE == 2, DevBO, DevB>::type;
  using BD = Body<BK, GEO, DT_BF16>;
  constexpr int NWAVE = MODE == 1 ? 16 : 8;
  BD::setup_tables(a.tab, a.t);
  fill_e<BK>(NWAVE * 64);
  __syncthreads();
  const int wv = BK::wave();
  typename BD::Unit un;
  un.eb = 0; un.wq = wv;
  typename BD::InnerRegs R;
  const int h = blockIdx.x % a.H;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if constexpr (MODE == 0) {
    BD::load_inner(R, un);
#pragma unroll 1
    for (int it = 0; it < iters; it++)
#pragma unroll 1
      for (int tt = 0; tt < 4; tt += 2) BD::template inner_tile2<false>(a, h, wv * 4 + tt, R, un);
  } else if constexpr (MODE == 3) {
    BD::load_inner(R, un);
#pragma unroll 1
    for (int it = 0; it < iters; it++)
#pragma unroll 1
      for (int tt = 0; tt < 4; tt++) {
        typename BD::KfRegs kf;
        BD::load_kf(a, h, wv * 4 + tt, kf);
        typename BD::A16 re, im;
        BD::template tile_fwd<true>(wv * 4 + tt, R, un, re, im);
        BD::kf_mul(a, kf, re, im);
        BD::template tile_inv<true, false>(a.s_inv, wv * 4 + tt, R, un, re, im);
      }
  } else {
    constexpr int TPWV = 32 / NWAVE;
    BD::template load_inner<false>(R, un);
#pragma unroll 1
    for (int it = 0; it < iters; it++)
#pragma unroll 1
      for (int tt = 0; tt < TPWV; tt++) {
        const int tau = wv * TPWV + tt;
        typename BD::KfRegs kf;
        BD::load_kf(a, h, tau, kf);
        typename BD::A16 re, im;
        BD::template tile_fwd<false>(tau, R, un, re, im);
        BD::kf_mul(a, kf, re, im);
        BD::template tile_inv<false, false>(a.s_inv, tau, R, un, re, im);
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * NWAVE + wv] = t1 - t0;
}

template <int MODE>
static void run(const char* name, ConvArgs a, int iters, unsigned long long* d_cyc, int num_cu, double ticks_to_cycles) {
  constexpr int NWAVE = MODE == 1 ? 16 : 8;
  auto kern = pb_kernel<MODE>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEO::LDS_BYTES));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(num_cu), dim3(NWAVE * 64), GEO::LDS_BYTES, 0, a, iters / 4 + 1, d_cyc);
  CHECK(hipDeviceSynchronize());
  double best_ms = 1e9, cyc_med = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(num_cu), dim3(NWAVE * 64), GEO::LDS_BYTES, 0, a, iters, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best_ms) {
      best_ms = ms;
      std::vector<unsigned long long> hcy(num_cu * NWAVE);
      CHECK(hipMemcpy(hcy.data(), d_cyc, hcy.size() * 8, hipMemcpyDeviceToHost));
      std::sort(hcy.begin(), hcy.end());
      cyc_med = (double)hcy[hcy.size() / 2];
    }
  }
  const double tiles_per_cu = 32.0 * iters;
  printf("%-58s %8.3f ms  %7.1f ns/tile/CU   %7.0f cycles/tile/SIMD (median wave, s_memtime)\n", name, best_ms, best_ms * 1e6 / tiles_per_cu,
         cyc_med * ticks_to_cycles / (tiles_per_cu / 4));
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int num_cu = prop.multiProcessorCount;
  const double ticks_to_cycles = argc > 1 ? atof(argv[1]) : 1.0;     // from pipe_probe's calibration line
  HostPlan hp;
  if (!build_plan(32768, DT_BF16, &hp)) { printf("plan failed\n"); return 1; }
  uint8_t* d_blob;
  CHECK(hipMalloc(&d_blob, hp.blob.size()));
  CHECK(hipMemcpy(d_blob, hp.blob.data(), hp.blob.size(), hipMemcpyHostToDevice));
  const int H = 64;
  std::vector<uint32_t> kf((size_t)H * 32768);
  for (size_t i = 0; i < kf.size(); i++) {
    double ph = (double)((i * 2654435761ull) % 100003) / 100003.0 * 6.283185307179586;
    kf[i] = f32_to_bf16((float)cos(ph)) | ((uint32_t)f32_to_bf16((float)sin(ph)) << 16);
  }
  uint32_t* d_kf;
  CHECK(hipMalloc(&d_kf, kf.size() * 4));
  CHECK(hipMemcpy(d_kf, kf.data(), kf.size() * 4, hipMemcpyHostToDevice));
  unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_cyc, 4096 * 16 * 8));
  ConvArgs a{};
  a.kf = d_kf; a.tab = d_blob; a.t = hp.tabs; a.H = H; a.B = 2; a.L = 16384; a.npair = 1;
  a.s_inv = 1.0f / 1024.0f; a.s_fwd = 1.0f; a.fast = 1;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  printf("phase-B probe, fft 32768 bf16, %d CUs, %d iterations x 32 tiles per CU%s\n", num_cu, iters,
#if defined(FFC_KO)
         "  [FFC_KO build: outer inverse twiddle removed]"
#else
         ""
#endif
  );
  run<0>("A  8 waves x 256 regs, 2 tiles in lock-step (shipped fwd)", a, iters, d_cyc, num_cu, ticks_to_cycles);
  run<3>("D  8 waves x 256 regs, 1 tile at a time, twiddle resident", a, iters, d_cyc, num_cu, ticks_to_cycles);
  run<2>("C  8 waves x 128 regs, 1 tile at a time, twiddle from LDS", a, iters, d_cyc, num_cu, ticks_to_cycles);
  run<1>("B 16 waves x 128 regs, 1 tile at a time, twiddle from LDS", a, iters, d_cyc, num_cu, ticks_to_cycles);
  return 0;
}
