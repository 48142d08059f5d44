// gfx950 VALU issue rates of the instructions the fused kernels could use for unpacking / merging 16-bit halves and for moving the dk_f
// sums, alone and beside MFMAs, at 1 / 2 / 4 waves per SIMD (round 6; successor of pipe_probe.hip's valu_only rows).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ uint8_t smem[];
enum { K_FMA, K_LSHL, K_AND, K_MULU24, K_LSHLOR, K_ANDOR, K_PACKF16, K_PACKF16HI, K_PERM, K_ALIGNBIT, K_BFI, K_MOVSDWA, K_ACCRD, K_ACCWR, K_CVTPK, K_CNDMASK, K_XOR, K_ADDU, K_MOV,
       K_FMAMIX, K_PKMUL, K_LSHLADD, K_MADU24, K_NKINDS };
static const char* NAMES[] = {"v_fma_f32", "v_lshlrev_b32 16", "v_and_b32 0xffff0000", "v_mul_u32_u24 0x10000", "v_lshl_or_b32", "v_and_or_b32", "v_pack_b32_f16", "v_pack_b32_f16 op_sel hi",
  "v_perm_b32", "v_alignbit_b32", "v_bfi_b32", "v_mov_b32_sdwa WORD_1<-WORD_0", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_cvt_pk_bf16_f32", "v_cndmask_b32", "v_xor_b32", "v_add_u32",
  "v_mov_b32", "v_fma_mix_f32", "v_pk_mul_f32", "v_lshl_add_u32", "v_mad_u32_u24"};
template <int K>
__device__ __forceinline__ void op(uint32_t& d, uint32_t a, uint32_t b, float& f, float g) {
  if (K == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(g));
  else if (K == K_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(d) : "v"(a));
  else if (K == K_AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(a));
  else if (K == K_MULU24) asm volatile("v_mul_u32_u24 %0, 0x10000, %1" : "=v"(d) : "v"(a));
  else if (K == K_LSHLOR) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_ANDOR) asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(d) : "v"(a), "v"(b), "s"(0xffffu));
  else if (K == K_PACKF16) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_PACKF16HI) asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(0x07060302u));
  else if (K == K_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_BFI) asm volatile("v_bfi_b32 %0, %3, %1, %2" : "=v"(d) : "v"(a), "v"(b), "s"(0xffffu));
  else if (K == K_MOVSDWA) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(d) : "v"(a));
  else if (K == K_ACCRD) asm volatile("v_accvgpr_read_b32 %0, a7" : "=v"(d));
  else if (K == K_ACCWR) asm volatile("v_accvgpr_write_b32 a9, %0" :: "v"(a));
  else if (K == K_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(f), "v"(g));
  else if (K == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_XOR) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_ADDU) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(a));
  else if (K == K_FMAMIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(f) : "v"(a), "v"(b));
  else if (K == K_LSHLADD) asm volatile("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(d) : "v"(a), "v"(b));
  else if (K == K_MADU24) asm volatile("v_mad_u32_u24 %0, %1, %3, %2" : "=v"(d) : "v"(a), "v"(b), "s"(0x10000u));
}
// NV instructions of kind K per MFMA (NM MFMAs per group of 8; NM = 0: VALU only)
template <int K, int NV, int NM>
__global__ __launch_bounds__(1024) void mix(unsigned long long* cyc, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  uint32_t d[8], a = 0x3f803f80u + lane, b = 0x40004000u + lane;
  float f[8], g = 1.00001f;
  for (int i = 0; i < 8; i++) { d[i] = 0; f[i] = 1.0f + i + lane * 1e-3f; }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 ma = {a, a, a, a}, mb = {b, b, b, b};
  asm volatile("v_accvgpr_write_b32 a7, %0\n v_accvgpr_write_b32 a9, %0" ::"v"(a) : "a7", "a9");
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int gI = 0; gI < 8; gI++) {
      if (NM) {
        if (gI & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(ma), "v"(mb));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ma), "v"(mb));
      }
#pragma unroll
      for (int v = 0; v < NV; v++) op<K>(d[(gI * NV + v) & 7], a, b, f[(gI * NV + v) & 7], g);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += f[i] + __builtin_bit_cast(float, d[i]);
  for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
static unsigned long long* d_cyc; static float* d_sink; static int num_cu;
template <int K, int NV, int NM>
static double run(int w, int iters) {
  auto kern = mix<K, NV, NM>;
  const int lds = 100 * 1024;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(num_cu), dim3(256 * w), lds, 0, d_cyc, d_sink, iters / 4);
  hipLaunchKernelGGL(kern, dim3(num_cu), dim3(256 * w), lds, 0, d_cyc, d_sink, iters);
  CHECK(hipDeviceSynchronize());
  const int nw = num_cu * w * 4;
  std::vector<unsigned long long> h(nw);
  CHECK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return (double)h[nw - 1 - nw / 50] / iters;
}
template <int K>
static void row() {
  printf("%-32s alone, cycles per instruction per SIMD:", NAMES[K]);
  for (int w : {1, 2, 4}) printf("  w%d %5.2f", w, run<K, 8, 0>(w, 1500) / 64.0 / w);
  printf("   | beside MFMAs (8 per MFMA), cycles per MFMA per SIMD:");
  for (int w : {2, 4}) printf("  w%d %5.1f", w, run<K, 8, 1>(w, 800) / 8.0 / w);
  printf("\n");
  if constexpr (K + 1 < K_NKINDS) row<K + 1>();
}
int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  num_cu = prop.multiProcessorCount;
  CHECK(hipMalloc(&d_cyc, 8 * 4096 * 8)); CHECK(hipMalloc(&d_sink, 64));
  printf("device %s, %d CUs (s_memtime ticks; MFMA-only stream = %.1f ticks per MFMA at 1 wave/SIMD)\n", prop.name, num_cu, run<K_FMA, 0, 1>(1, 2000) / 8.0);
  row<0>();
  return 0;
}
