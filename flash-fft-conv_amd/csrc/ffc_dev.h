// gfx950 wave backend + launch helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#define FFC_FN __device__ __forceinline__
#include "../../include/flashfftconv_hip.h"
#include "ffc_body.h"
#include "ffc_modes.h"
#include "ffc_big.h"

namespace ffc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) uint8_t ffc_smem[];

// gfx950 wave backend: every "vector" of the body is one value per lane.
struct DevB {
  using f32 = float;
  using i32 = int;
  using u32 = uint32_t;
  using pred = bool;
  struct U2 { u32 x, y; };
  struct U4 { u32 x, y, z, w; };
  using A16 = f32x16;   // 16 consecutive VGPRs/AGPRs: the MFMA accumulator tuple
  using W4 = u32x4v;    // 4 consecutive VGPRs: one MFMA A/B operand
  static constexpr bool HAS_TR = true;
  // element-wise complex multiply of two accumulator tuples (x (x) t or x (x) conj t): whole-vector fp32 ops,
  // which gfx950 legalises to v_pk_mul_f32 / v_pk_fma_f32 on aligned register pairs
  template <bool CONJ> static FFC_FN void cmul16(A16& re, A16& im, const A16& tr, const A16& ti) {
    const A16 a = re, b = im;
    if (!CONJ) { re = a * tr - b * ti; im = a * ti + b * tr; }
    else { re = a * tr + b * ti; im = b * tr - a * ti; }
  }
  static FFC_FN A16 a16_scale(const A16& a, float s) { return a * s; }
  static FFC_FN A16 a16_zero() { A16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; return z; }
  static FFC_FN W4 w4(u32 a, u32 b, u32 c, u32 e) { W4 v = {a, b, c, e}; return v; }

  static FFC_FN i32 lane() { return (int)(threadIdx.x & 63); }
  static FFC_FN int wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
  static FFC_FN void barrier() { __syncthreads(); }
  static FFC_FN f32 fconst(float c) { return c; }
  static FFC_FN pred ptrue() { return true; }
  static FFC_FN pred pfalse() { return false; }
  static FFC_FN f32 as_f32(u32 a) { return __builtin_bit_cast(float, a); }
  static FFC_FN u32 as_u32(f32 a) { return __builtin_bit_cast(uint32_t, a); }

  static FFC_FN U2 lds_r64(i32 off) { uint2 v = *(const uint2*)(ffc_smem + off); return U2{v.x, v.y}; }
  static FFC_FN void lds_w64(i32 off, U2 v) { *(uint2*)(ffc_smem + off) = make_uint2(v.x, v.y); }
  static FFC_FN void lds_w32(i32 off, u32 v) { *(uint32_t*)(ffc_smem + off) = v; }
  static FFC_FN void lds_w128(i32 off, U4 v, pred p) {
    if (p) *(uint4*)(ffc_smem + off) = make_uint4(v.x, v.y, v.z, v.w);
  }
  static FFC_FN U4 lds_r128(i32 off) { uint4 v = *(const uint4*)(ffc_smem + off); return U4{v.x, v.y, v.z, v.w}; }
  static FFC_FN u32 lds_r32(i32 off) { return *(const uint32_t*)(ffc_smem + off); }
  static FFC_FN u32 lds_r16(i32 off) { return *(const uint16_t*)(ffc_smem + off); }
  static FFC_FN U2 lds_r64_tr(i32 off) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ffc_smem + off));
    uint2 v = __builtin_bit_cast(uint2, t);
    return U2{v.x, v.y};
  }
  static FFC_FN void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  static FFC_FN u32 uconst(uint32_t c) { return c; }
  static FFC_FN i32 mul24(i32 a, i32 b) { return __mul24(a, b); }
  static FFC_FN unsigned long long clock() { return __builtin_amdgcn_s_memtime(); }
  static FFC_FN U2 u2_from64(unsigned long long v) { return U2{(u32)v, (u32)(v >> 32)}; }
  // v_sin/v_cos run on the transcendental unit; consumers scheduled right behind them (packed f32 math in
  // particular) were observed to read stale operands on gfx950 (timing-dependent 1-3% errors, caught by a
  // run-to-run determinism check).  An opaque asm with wait states orders them conservatively.
  static FFC_FN void settle(f32& a, f32& b) { asm volatile("s_nop 4" : "+v"(a), "+v"(b)); }
  static FFC_FN f32 i2f(i32 a) { return (float)a; }
  static FFC_FN f32 cos_rev(f32 x) { return __builtin_amdgcn_cosf(x); }   // v_cos_f32: argument in revolutions
  static FFC_FN f32 sin_rev(f32 x) { return __builtin_amdgcn_sinf(x); }
  // hide a value from LICM/CSE so per-phase address math is recomputed instead of kept live
  static FFC_FN i32 opaque(i32 x) { asm volatile("" : "+v"(x)); return x; }
  static FFC_FN u32 sel(pred p, u32 a, u32 b) { return p ? a : b; }
  static FFC_FN u32 g_r16(const void* base, i32 e, pred p) {
    uint16_t v = 0;
    if (p) v = ((const uint16_t*)base)[e];
    return v;
  }
  static FFC_FN void g_w16(void* base, i32 e, u32 v, pred p) {
    if (p) ((uint16_t*)base)[e] = (uint16_t)v;
  }
  static FFC_FN u32 g_r32(const void* base, i32 e, pred p) {
    uint32_t v = 0;
    if (p) v = ((const uint32_t*)base)[e];
    return v;
  }
  static FFC_FN void g_w32(void* base, i32 e, u32 v, pred p) {
    if (p) ((uint32_t*)base)[e] = v;
  }
  static FFC_FN U2 g_r64(const void* base, i32 o8, pred p) {
    uint2 v = make_uint2(0, 0);
    if (p) v = ((const uint2*)base)[o8];
    return U2{v.x, v.y};
  }
  static FFC_FN void g_w64(void* base, i32 o8, U2 v, pred p) {
    if (p) ((uint2*)base)[o8] = make_uint2(v.x, v.y);
  }
  static FFC_FN U4 g_r128(const void* base, i32 o16) {
    uint4 v = ((const uint4*)base)[o16];
    return U4{v.x, v.y, v.z, v.w};
  }
  static FFC_FN U4 g_r128p(const void* base, i32 o16, pred p) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p) v = ((const uint4*)base)[o16];
    return U4{v.x, v.y, v.z, v.w};
  }
  static FFC_FN void g_w128(void* base, i32 o16, U4 v, pred p) {
    if (p) ((uint4*)base)[o16] = make_uint4(v.x, v.y, v.z, v.w);
  }
  template <int DT>
  static FFC_FN void mfma(A16& acc, const W4& a, const W4& b) {
    if (DT == DT_BF16)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    else
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
  template <int DT>
  static FFC_FN u32 pack(f32 lo, f32 hi) {
    f32x2 v = {lo, hi};
    if (DT == DT_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  template <int DT>
  static FFC_FN f32 unpack_lo(u32 a) {
    if (DT == DT_BF16) return __builtin_bit_cast(float, a << 16);
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(a & 0xffffu));
  }
  template <int DT>
  static FFC_FN f32 unpack_hi(u32 a) {
    if (DT == DT_BF16) return __builtin_bit_cast(float, a & 0xffff0000u);
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(a >> 16));
  }
};

// blockIdx -> (head, chunk).  Blocks land on XCD (id % 8): keep all chunks of one head on one XCD
// so k_f[h] is served by that XCD's L2 (speed only, never correctness).
__device__ __forceinline__ bool map_block(int H, int nchunk, int* h, int* chunk) {
  int id = blockIdx.x;
  int xcd = id & 7, s = id >> 3;
  *h = xcd + 8 * (s / nchunk);
  *chunk = s % nchunk;
  return *h < H;
}

}  // namespace ffc

struct ffc_plan {
  ffc::HostPlan hp;
  ffc::HostPlan hp_bf;          // bf16 tables for the dk inverse (fp32 dynamic range), == hp for bf16 plans
  uint8_t* d_blob = nullptr;
  uint8_t* d_blob_bf = nullptr;
  int32_t* d_freq = nullptr;
  int num_cu = 256;
};

extern "C" void ffc_set_error_(const char* m);
static inline int ffc_fail(const std::string& m) { ffc_set_error_(m.c_str()); return 1; }
#define HIPCHK(x)                                                                     \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) return ffc_fail(std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class K>
static int ffc_set_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return ffc_fail(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
  return 0;
}

template <template <class, int> class FN, class... A>
static int ffc_dispatch(int N, int dtype, A&&... args) {
  using namespace ffc;
#define FFC_CASE(NN, a, b, c) \
  case NN:                    \
    return dtype == DT_BF16 ? FN<Geo<a, b, c>, DT_BF16>::run(args...) : FN<Geo<a, b, c>, DT_F16>::run(args...);
  switch (N) {
    FFC_CASE(256, 1, 16, 16)
    FFC_CASE(512, 1, 16, 32)
    FFC_CASE(1024, 1, 32, 32)
    FFC_CASE(4096, 16, 16, 16)
    FFC_CASE(8192, 32, 16, 16)
    FFC_CASE(16384, 16, 32, 32)
    FFC_CASE(32768, 32, 32, 32)
  }
#undef FFC_CASE
  return ffc_fail("unsupported fft size");
}

// pairs per chunk / number of chunks so the grid fills the chip (>= ~2 waves of workgroups) while a
// workgroup still loops over several pairs of one head (k_f[h] reuse through L2).
static inline void ffc_choose_chunks(const ffc_plan* p, int H, int npair, int* nchunk, int* ppc) {
  const bool outer = p->hp.N1 > 1;
  int upw = 8 / p->hp.NW;                       // units a workgroup processes per iteration
  int pairs_per_iter = outer ? upw : upw * p->hp.G;
  int wg_per_cu = outer ? 1 : 2;
  int mult = 2;   // see profiles/: larger values did not help (k_f re-reads are not the limiter)
  if (const char* e = getenv("FFC_WG_MULT")) mult = atoi(e) > 0 ? atoi(e) : 2;   // tuning knob
  int target = p->num_cu * wg_per_cu * mult;
  int iters_total = (npair + pairs_per_iter - 1) / pairs_per_iter;
  int nc = (target + H - 1) / H;
  if (nc > iters_total) nc = iters_total;
  if (nc < 1) nc = 1;
  int ipc = (iters_total + nc - 1) / nc;
  *ppc = ipc * pairs_per_iter;
  *nchunk = (npair + *ppc - 1) / *ppc;
}
