// gfx950 issue / pipe microbenchmark (standalone: hipcc --offload-arch=gfx950 -O3 pipe_probe.hip -o pipe_probe).
// Answers the question the fused FFT-convolution kernels hinge on: how do VALU instructions of the waves that share a
// SIMD interleave with each other and with the MFMAs of the same SIMD?
//   * cycles per instruction of v_fma_f32 / v_pk_fma_f32 / v_cvt_pk_bf16_f32 / v_mfma_f32_32x32x16_bf16 for
//     1, 2 and 4 waves per SIMD (one workgroup per CU, forced by its LDS request)
//   * cycles per MFMA of a stream of one MFMA + NV VALU instructions (NV = 4 .. 16: the fused kernels run 11 (phase B)
//     to 17 (whole kernel) VALU per MFMA), same occupancies
//   * also: measured peaks (stream copy GB/s, dense MFMA TFLOP/s) for bench.py's `peak_measured`
// Per-wave cycles come from s_memtime around the loop (shader clock), wall time from HIP events.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

extern __shared__ uint8_t smem[];

enum { K_FMA = 0, K_PKFMA = 1, K_CVT = 2, K_PKMUL = 3, K_MUL = 4, K_PERM = 5, K_DOT2 = 6, K_PKF16 = 7, K_CVTF16 = 8, K_ACCRD = 9, K_LSHL = 10, K_SIN = 11, K_CVTF32BF = 12, K_BITOP3 = 13 };

template <int KIND>
__device__ __forceinline__ void valu_op(f32x2& r, f32x2 c, f32x2 d, uint32_t& u) {
  if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x) : "v"(c.x), "v"(d.x));
  else if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r.x) : "v"(c.x));
  else if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
  else if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(c));
  else if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u) : "v"(r.x), "v"(r.y), "s"(0x07060302u));
  else if (KIND == K_DOT2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(r.x) : "v"(c.x), "v"(d.x));
  else if (KIND == K_PKF16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r.x) : "v"(c.x), "v"(d.x));
  else if (KIND == K_CVTF16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(r.x), "v"(r.y));
  else if (KIND == K_ACCRD) asm volatile("v_accvgpr_read_b32 %0, a7" : "=v"(u));
  else if (KIND == K_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u) : "v"(r.x));
  else if (KIND == K_SIN) asm volatile("v_sin_f32 %0, %1" : "=v"(u) : "v"(r.x));
  else if (KIND == K_CVTF32BF) asm volatile("v_cvt_f32_bf16 %0, %1" : "=v"(u) : "v"(r.x));
  else if (KIND == K_BITOP3) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x11" : "=v"(u) : "v"(r.x), "v"(r.y), "v"(c.x));
  else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(r.x), "v"(r.y));
}

// NV VALU instructions of KIND per MFMA (NM = 0: VALU only, 64 per iteration; NV = 0: MFMA only)
template <int KIND, int NV, int MFMA>
__global__ __launch_bounds__(1024) void mix_kernel(unsigned long long* cyc, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f32x2 r[8];
  uint32_t u[8];
  for (int i = 0; i < 8; i++) { r[i].x = 1.0f + lane * 1e-3f + i; r[i].y = 0.5f + i; u[i] = 0; }
  f32x2 c = {0.99991f, 1.00003f}, d = {1e-6f, -1e-6f};
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u};
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
  if (MFMA == 3) asm volatile(".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31\n v_accvgpr_write_b32 a\\r, 0\n .endr" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (MFMA == 1) {
        if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      } else if (MFMA == 2) {
        if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc0) : "v"(a), "v"(b));
      } else if (MFMA == 3) {
        if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(a), "v"(b) : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a), "v"(b) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
      } else if (MFMA == 4) {
        if (g & 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %3, %1, %2, %3" : "+v"(q1), "+v"(q3) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %3, %1, %2, %3" : "+v"(q0), "+v"(q2) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int v = 0; v < NV; v++) valu_op<KIND>(r[(g * NV + v) & 7], c, d, u[(g * NV + v) & 7]);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += r[i].x + r[i].y + __builtin_bit_cast(float, u[i]);
  for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
  for (int i = 0; i < 4; i++) s += q0[i] + q1[i] + q2[i] + q3[i];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

struct Res { double cyc_per_iter, ms; };

template <int KIND, int NV, int MFMA>
static Res run_mix(int waves_per_simd, int iters, unsigned long long* d_cyc, float* d_sink, int num_cu) {
  const int threads = 256 * waves_per_simd;
  const int lds = 100 * 1024;      // one workgroup per CU
  auto kern = mix_kernel<KIND, NV, MFMA>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(num_cu), dim3(threads), lds, 0, d_cyc, d_sink, iters / 4);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(num_cu), dim3(threads), lds, 0, d_cyc, d_sink, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = num_cu * waves_per_simd * 4;
  std::vector<unsigned long long> h(nw);
  CHECK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  Res r;
  r.cyc_per_iter = (double)h[nw - 1 - nw / 50] / iters;     // ~last-finishing wave (98th percentile): the SIMD's total
  r.ms = ms;
  return r;
}

static const char* kname(int k) {
  switch (k) { case K_FMA: return "v_fma_f32"; case K_PKFMA: return "v_pk_fma_f32"; case K_CVT: return "v_cvt_pk_bf16_f32"; case K_PKMUL: return "v_pk_mul_f32";
    case K_PERM: return "v_perm_b32"; case K_DOT2: return "v_dot2_f32_bf16"; case K_PKF16: return "v_pk_fma_f16"; case K_CVTF16: return "v_cvt_pk_f16_f32";
    case K_ACCRD: return "v_accvgpr_read_b32"; case K_LSHL: return "v_lshlrev_b32"; case K_SIN: return "v_sin_f32"; case K_CVTF32BF: return "v_cvt_f32_bf16"; case K_BITOP3: return "v_bitop3_b32";
    default: return "v_mul_f32"; }
}

template <int KIND>
static void valu_only(unsigned long long* d_cyc, float* d_sink, int num_cu, double clk_ratio) {
  for (int w : {1, 2, 4}) {
    Res r = run_mix<KIND, 8, 0>(w, 2000, d_cyc, d_sink, num_cu);
    // 64 instructions per iteration per wave
    printf("valu_only  %-18s waves/SIMD %d : %6.2f cyc/instr/wave  -> SIMD issues one every %5.2f cyc   (%.3f ms)\n", kname(KIND), w,
           r.cyc_per_iter * clk_ratio / 64.0, r.cyc_per_iter * clk_ratio / 64.0 / w, r.ms);
  }
}

template <int KIND, int NV, int MF = 1>
static void mix_row(unsigned long long* d_cyc, float* d_sink, int num_cu, double clk_ratio) {
  printf("mfma%s+%2d %-18s:", MF == 1 ? "(C=acc,vgpr)" : MF == 2 ? "(C=0,vgpr)  " : MF == 3 ? "(C=acc,agpr)" : "(2x16x16x32)", NV, kname(KIND));
  for (int w : {1, 2, 4}) {
    Res r = run_mix<KIND, NV, MF>(w, 1000, d_cyc, d_sink, num_cu);
    // 8 MFMA per iteration per wave; per SIMD w waves
    printf("   w%d %6.1f cyc/MFMA/SIMD (%5.2f ns)", w, r.cyc_per_iter * clk_ratio / 8.0 / w, r.ms * 1e6 / 1000 / 8.0 / w);
  }
  printf("\n");
}


// Phase-structured stream, as the fused kernels' tile chains: NM MFMAs (two accumulators, alternating), then NV VALU
// instructions.  DEP: the VALU block starts by reading the accumulators and the MFMA operand is rewritten at its end
// (the MFMA -> twiddle -> cvt -> MFMA chain of one tile); otherwise the two blocks are independent.
template <int KIND, int NM, int NV, bool DEP>
__global__ __launch_bounds__(1024) void block_kernel(unsigned long long* cyc, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f32x2 r[8];
  uint32_t u[8];
  for (int i = 0; i < 8; i++) { r[i].x = 1.0f + lane * 1e-3f + i; r[i].y = 0.5f + i; u[i] = 0; }
  f32x2 c = {0.99991f, 1.00003f}, d = {1e-6f, -1e-6f};
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < NM; g++) {
      if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
    }
    if (DEP) {
      asm volatile("s_nop 15\n s_nop 3\n v_fma_f32 %0, %1, %2, %0" : "+v"(r[0].x) : "v"(acc0[15]), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[1].x) : "v"(acc1[15]), "v"(c.x));
    }
#pragma unroll
    for (int v = 0; v < NV; v++) valu_op<KIND>(r[v & 7], c, d, u[v & 7]);
    if (DEP) {
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a.x) : "v"(r[0].x), "v"(r[1].x));
      acc0[0] = 0.f; acc1[0] = 0.f;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += r[i].x + r[i].y + __builtin_bit_cast(float, u[i]);
  for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NM, int NV, bool DEP>
static void block_row(unsigned long long* d_cyc, float* d_sink, int num_cu) {
  printf("block %d MFMA + %3d %-18s %s:", NM, NV, kname(KIND), DEP ? "dependent  " : "independent");
  auto kern = block_kernel<KIND, NM, NV, DEP>;
  const int lds = 100 * 1024, iters = 400;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w : {1, 2, 3, 4}) {
    hipLaunchKernelGGL(kern, dim3(num_cu), dim3(256 * w), lds, 0, d_cyc, d_sink, iters / 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(num_cu), dim3(256 * w), lds, 0, d_cyc, d_sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = num_cu * w * 4;
    std::vector<unsigned long long> h(nw);
    CHECK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("  w%d %6.0f cyc/block/SIMD (%5.0f ns)", w, (double)h[nw - 1 - nw / 50] / iters / w, ms * 1e6 / iters / w);
  }
  printf("\n");
}

// ---- peaks
__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
  if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int num_cu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, num_cu, prop.clockRate);
  unsigned long long* d_cyc; float* d_sink;
  CHECK(hipMalloc(&d_cyc, 8 * 4096 * 8)); CHECK(hipMalloc(&d_sink, 64));
  // calibrate s_memtime ticks against wall time with an MFMA-only stream (32 cycles per MFMA per SIMD, documented)
  Res cal = run_mix<K_FMA, 0, 1>(1, 4000, d_cyc, d_sink, num_cu);
  printf("calibration: MFMA-only, 1 wave/SIMD: %.2f s_memtime ticks per MFMA, %.3f ms for %d MFMA -> %.2f ns per MFMA\n", cal.cyc_per_iter / 8.0, cal.ms,
         4000 * 8, cal.ms * 1e6 / (4000 * 8));
  // ticks -> shader cycles: if an MFMA is 32 shader cycles, ratio = 32 / ticks_per_mfma
  double clk_ratio = 32.0 / (cal.cyc_per_iter / 8.0);
  printf("assuming 32 shader cycles per back-to-back MFMA: 1 tick = %.3f cycles, shader clock %.0f MHz\n", clk_ratio, 32.0 / (cal.ms * 1e6 / (4000 * 8)) * 1e3);
  for (int w : {1, 2, 4}) {
    Res r = run_mix<K_FMA, 0, 1>(w, 2000, d_cyc, d_sink, num_cu);
    printf("mfma_only  waves/SIMD %d : %6.1f cyc/MFMA/SIMD\n", w, r.cyc_per_iter * clk_ratio / 8.0 / w);
  }
  valu_only<K_FMA>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_MUL>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_PKFMA>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_PKMUL>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_CVT>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_PERM>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_DOT2>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_PKF16>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_CVTF16>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_ACCRD>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_LSHL>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_SIN>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_CVTF32BF>(d_cyc, d_sink, num_cu, clk_ratio);
  valu_only<K_BITOP3>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 2>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 4>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 6>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 8>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 11>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 16>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 8, 2>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 11, 2>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 16, 2>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 8, 3>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 11, 3>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 16, 3>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 0, 4>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 8, 4>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 11, 4>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_FMA, 16, 4>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_PERM, 11>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_PKF16, 11>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_CVTF16, 8>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_DOT2, 8>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_PKFMA, 4>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_PKFMA, 8>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_CVT, 8>(d_cyc, d_sink, num_cu, clk_ratio);
  mix_row<K_CVT, 16>(d_cyc, d_sink, num_cu, clk_ratio);

  // realistic phase-B proportions per 8-MFMA block: ~90 VALU
  block_row<K_FMA, 8, 90, false>(d_cyc, d_sink, num_cu);
  block_row<K_FMA, 8, 90, true>(d_cyc, d_sink, num_cu);
  block_row<K_FMA, 8, 60, true>(d_cyc, d_sink, num_cu);
  block_row<K_FMA, 8, 45, true>(d_cyc, d_sink, num_cu);
  block_row<K_PKFMA, 8, 45, true>(d_cyc, d_sink, num_cu);
  block_row<K_CVT, 8, 45, true>(d_cyc, d_sink, num_cu);
  block_row<K_FMA, 16, 180, true>(d_cyc, d_sink, num_cu);
  // peaks
  {
    size_t bytes = (size_t)2 << 30;
    uint4 *s, *d;
    CHECK(hipMalloc(&s, bytes)); CHECK(hipMalloc(&d, bytes));
    CHECK(hipMemset(s, 1, bytes)); CHECK(hipMemset(d, 2, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 0;
    for (int rep = 0; rep < 5; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(copy_kernel, dim3(num_cu * 8), dim3(512), 0, 0, s, d, bytes / 16);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = std::max(best, 2.0 * bytes / (ms * 1e-3) / 1e9);
    }
    printf("peak_measured stream copy (2 GiB read + 2 GiB write, 16 B per lane): %.0f GB/s\n", best);
    int iters = 20000;
    best = 0;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(mfma_peak_kernel, dim3(num_cu * 4), dim3(256), 0, 0, d_sink, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)num_cu * 4 * 4 * iters * 4 * (2.0 * 32 * 32 * 16);
      best = std::max(best, flops / (ms * 1e-3) / 1e12);
    }
    printf("peak_measured v_mfma_f32_32x32x16_bf16 dense: %.0f TFLOP/s\n", best);
  }
  return 0;
}
