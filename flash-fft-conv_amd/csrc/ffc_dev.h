// gfx950 wave backend + launch helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>

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
  static constexpr bool LEAN_OUTER = false;
  static constexpr bool FAST_ONLY = false;     // true: only the 16-byte arm of the row accesses is compiled (Body::gload8 / gstore8)
  using A16 = f32x16;   // 16 consecutive VGPRs/AGPRs: the MFMA accumulator tuple
  using W4 = u32x4v;    // 4 consecutive VGPRs: one MFMA A/B operand
  static constexpr bool HAS_TR = true;
  // element-wise complex multiply of two accumulator tuples (x (x) t or x (x) conj t): whole-vector fp32 ops,
  // which gfx950 legalises to v_pk_mul_f32 / v_pk_fma_f32 on aligned register pairs
#ifndef FFC_NO_PK
#define FFC_NO_PK 0
#endif
#if !FFC_NO_PK
  template <bool CONJ> static FFC_FN void cmul16(A16& re, A16& im, const A16& tr, const A16& ti) {
    const A16 a = re, b = im;
    if (!CONJ) { re = a * tr - b * ti; im = a * ti + b * tr; }
    else { re = a * tr + b * ti; im = b * tr - a * ti; }
  }
  using F2 = f32x2;
  static FFC_FN F2 f2(f32 a, f32 b) { F2 v = {a, b}; return v; }
  static FFC_FN f32 f2_lo(F2 v) { return v.x; }
  static FFC_FN f32 f2_hi(F2 v) { return v.y; }
  // (yr, yi) = (xr, xi) * (wr + i wi) on a pair of complex values
  static FFC_FN void cmulp(F2 xr, F2 xi, f32 wr, f32 wi, F2& yr, F2& yi) {
    yr = xr * wr - xi * wi;
    yi = xr * wi + xi * wr;
  }
  template <bool CONJ>
  static FFC_FN void cmul2v(A16& re, A16& im, int r0, F2 tr, F2 ti) {
    const F2 a = {re[r0], re[r0 + 1]}, b = {im[r0], im[r0 + 1]};
    F2 x, y;
    if (!CONJ) { x = a * tr - b * ti; y = a * ti + b * tr; }
    else { x = a * tr + b * ti; y = b * tr - a * ti; }
    re[r0] = x.x; re[r0 + 1] = x.y; im[r0] = y.x; im[r0 + 1] = y.y;
  }
  // (wr, wi) += x[r0..r0+1] (x) conj z : packed fp32 math on register pairs
  static FFC_FN void cmac2_conj(F2& wr, F2& wi, const A16& a, const A16& b, int r0, F2 zr, F2 zi) {
    const F2 a2 = {a[r0], a[r0 + 1]}, b2 = {b[r0], b[r0 + 1]};
    wr = wr + (a2 * zr + b2 * zi);
    wi = wi + (b2 * zr - a2 * zi);
  }
#else
  // scalar fp32 variant of the same helpers (A/B of packed vs plain VALU next to MFMAs: MI355X_MICROARCH.md prices a
  // v_pk_fma_f32 beside MFMAs well above two v_fma_f32)
  static FFC_FN f32 sfma(f32 a, f32 b, f32 c) { return __builtin_fmaf(a, b, c); }
  template <bool CONJ> static FFC_FN void cmul16(A16& re, A16& im, const A16& tr, const A16& ti) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const f32 a = re[r], b = im[r];
      if (!CONJ) { re[r] = sfma(a, tr[r], -(b * ti[r])); im[r] = sfma(a, ti[r], b * tr[r]); }
      else { re[r] = sfma(a, tr[r], b * ti[r]); im[r] = sfma(b, tr[r], -(a * ti[r])); }
    }
  }
  struct F2 { f32 x, y; };
  static FFC_FN F2 f2(f32 a, f32 b) { F2 v; v.x = a; v.y = b; return v; }
  static FFC_FN f32 f2_lo(F2 v) { return v.x; }
  static FFC_FN f32 f2_hi(F2 v) { return v.y; }
  static FFC_FN void cmulp(F2 xr, F2 xi, f32 wr, f32 wi, F2& yr, F2& yi) {
    yr.x = sfma(xr.x, wr, -(xi.x * wi)); yr.y = sfma(xr.y, wr, -(xi.y * wi));
    yi.x = sfma(xr.x, wi, xi.x * wr); yi.y = sfma(xr.y, wi, xi.y * wr);
  }
  template <bool CONJ>
  static FFC_FN void cmul2v(A16& re, A16& im, int r0, F2 tr, F2 ti) {
    const f32 t_r[2] = {tr.x, tr.y}, t_i[2] = {ti.x, ti.y};
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const f32 a = re[r0 + q], b = im[r0 + q];
      if (!CONJ) { re[r0 + q] = sfma(a, t_r[q], -(b * t_i[q])); im[r0 + q] = sfma(a, t_i[q], b * t_r[q]); }
      else { re[r0 + q] = sfma(a, t_r[q], b * t_i[q]); im[r0 + q] = sfma(b, t_r[q], -(a * t_i[q])); }
    }
  }
  static FFC_FN void cmac2_conj(F2& wr, F2& wi, const A16& a, const A16& b, int r0, F2 zr, F2 zi) {
    wr.x = wr.x + sfma(a[r0], zr.x, b[r0] * zi.x); wr.y = wr.y + sfma(a[r0 + 1], zr.y, b[r0 + 1] * zi.y);
    wi.x = wi.x + sfma(b[r0], zr.x, -(a[r0] * zi.x)); wi.y = wi.y + sfma(b[r0 + 1], zr.y, -(a[r0 + 1] * zi.y));
  }
#endif
  // Accumulation registers a0..a127 addressed by number (see Modes::WAcc).  The kernel marks them used once
  // (agpr_reserve) so that the kernel descriptor allocates them; the compiler itself never places values there
  // (MFMAs are kept in VGPR form, build flag -mllvm --amdgpu-mfma-vgpr-form; build.py checks the disassembly).
#define FFC_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
  static FFC_FN void agpr_reserve() {
    asm volatile("; a0..a127 hold the dk_f partial sums"
                 ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", FFC_A8(1), FFC_A8(2), FFC_A8(3), FFC_A8(4),
                     FFC_A8(5), FFC_A8(6), FFC_A8(7), FFC_A8(8), FFC_A8(9), FFC_A8(10), FFC_A8(11), "a120", "a121", "a122",
                     "a123", "a124", "a125", "a126", "a127");
  }
#undef FFC_A8
  template <int I> static FFC_FN f32 agpr_get() {
    f32 x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(I));
    return x;
  }
  template <int I> static FFC_FN void agpr_set(f32 x) { asm volatile("v_accvgpr_write_b32 a%c0, %1" ::"n"(I), "v"(x)); }
  // Round 6: acc[I0 .. I0+15] (accumulation registers) += A x B, bf16 operands.  The dk_f sums are accumulated by the matrix pipe (Modes::
  // w_acc_tile: B = the products D (x) conj Z rounded to bf16, A = a permuted identity), so the VALU never shuttles them through
  // v_accvgpr_read / v_accvgpr_write.  Inline asm (the compiler keeps its own MFMAs in VGPR form and must not see a0..a127): the wait states
  // are ours (cdna_hip_programming.md 5.7 item 2) -- `s_nop 1` covers the VALU-written A / B operands; an MFMA that takes the previous
  // one's destination whole as C needs none; readers of the sums call mfma_settle() first.
  template <int I0> static FFC_FN void mfma_acc_bf16(const W4& a, const W4& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(I0), "n"(I0 + 15));
  }
  // an MFMA's destination -> any reader other than the next accumulating MFMA: 12 wait states for the 8-pass 32x32x16 (ibid.)
  static FFC_FN void mfma_settle() { asm volatile("s_nop 15"); }
  // keep a load-defined MFMA operand in architectural VGPRs (the allocator may otherwise place it in the
  // accumulation registers, which hold the dk_f partial sums in the backward kernels)
  static FFC_FN void pin(W4& x) { asm("" : "+v"(x)); }
  static FFC_FN void pin4(U4& x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }
  static FFC_FN void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
  // rows r0, r0+1 of (re,im) times (t0,t1) (or its conjugate): one packed mul + one packed fma per output pair
  template <bool CONJ>
  static FFC_FN void cmul2(A16& re, A16& im, int r0, f32 tr0, f32 tr1, f32 ti0, f32 ti1) {
    cmul2v<CONJ>(re, im, r0, f2(tr0, tr1), f2(ti0, ti1));
  }
#if !FFC_NO_PK
  static FFC_FN A16 a16_scale(const A16& a, float s) { return a * s; }
#else
  static FFC_FN A16 a16_scale(const A16& a, float s) {
    A16 o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = a[r] * s;
    return o;
  }
#endif
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
  static FFC_FN void lds_w16(i32 off, u32 v) { *(uint16_t*)(ffc_smem + off) = (uint16_t)v; }
  static FFC_FN u32 lds_r32(i32 off) { return *(const uint32_t*)(ffc_smem + off); }
  static FFC_FN u32 lds_r16(i32 off) { return *(const uint16_t*)(ffc_smem + off); }
  static FFC_FN U2 lds_r64_tr(i32 off) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ffc_smem + off));
    uint2 v = __builtin_bit_cast(uint2, t);
    return U2{v.x, v.y};
  }
  static FFC_FN void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  // LDS-DMA (global_load_lds_dword): lane's dword `dw` of base -> LDS[lds_off + 4 * lane]; lds_off is wave-uniform (M0).
  // No VGPR destination; completion is counted by vmcnt (vm_wait0).  nt: streaming policy (rows read once).
  template <bool NT>
  static FFC_FN void g2lds32(const void* base, i32 dw, int lds_off) {
    const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)((const uint32_t*)base + dw);
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(ffc_smem + lds_off);
    if constexpr (NT) __builtin_amdgcn_global_load_lds(g, l, 4, 0, 2);
    else __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
  }
  // 16-byte form: lane's 16 bytes at base[o16] -> LDS[lds_off + 16 * lane] (1 KB per wave instruction)
  template <bool NT>
  static FFC_FN void g2lds128(const void* base, i32 o16, int lds_off) {
    const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)((const uint4*)base + o16);
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(ffc_smem + lds_off);
    if constexpr (NT) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 2);
    else __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
  }
  static FFC_FN void vm_wait0() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // s_waitcnt vmcnt(0)
  static FFC_FN void lds_w32p(i32 off, u32 v, pred p) { if (p) *(uint32_t*)(ffc_smem + off) = v; }
  static FFC_FN pred pnot(pred p) { return !p; }
  static FFC_FN u32 uconst(uint32_t c) { return c; }
  static FFC_FN i32 mul24(i32 a, i32 b) { return __mul24(a, b); }
  static FFC_FN unsigned long long clock() { return __builtin_amdgcn_s_memtime(); }
  // wave priority (s_setprio): see Body::outer_jobs
  template <int P> static FFC_FN void setprio() { __builtin_amdgcn_s_setprio(P); }
  static FFC_FN U2 u2_from64(unsigned long long v) { return U2{(u32)v, (u32)(v >> 32)}; }
  // v_sin/v_cos run on the transcendental unit; consumers scheduled right behind them (packed f32 math in
  // particular) were observed to read stale operands on gfx950 (timing-dependent 1-3% errors, caught by a
  // run-to-run determinism check).  An opaque asm with wait states orders them conservatively.
#ifndef FFC_NO_SETTLE
  static FFC_FN void settle(f32& a, f32& b) { asm volatile("s_nop 4" : "+v"(a), "+v"(b)); }
#else
  static FFC_FN void settle(f32&, f32&) {}      // hazard experiment (benchmarks/hazard_probe.py): no wait states
#endif
  static FFC_FN f32 i2f(i32 a) { return (float)a; }
  static FFC_FN f32 cos_rev(f32 x) { return __builtin_amdgcn_cosf(x); }   // v_cos_f32: argument in revolutions
  static FFC_FN f32 sin_rev(f32 x) { return __builtin_amdgcn_sinf(x); }
  // hide a value from LICM/CSE so per-phase address math is recomputed instead of kept live
  static FFC_FN i32 opaque(i32 x) { asm volatile("" : "+v"(x)); return x; }
  static FFC_FN u32 sel(pred p, u32 a, u32 b) { return p ? a : b; }
  // 16-bit half merges in one v_perm_b32: (lo16(a) | lo16(b) << 16) and (hi16(a) | hi16(b) << 16)
  static FFC_FN u32 merge_lo(u32 a, u32 b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
  static FFC_FN u32 merge_hi(u32 a, u32 b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
  static FFC_FN i32 imin(i32 a, int b) { return a < b ? a : b; }
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
  // streaming variants (activation rows: read once / written once): non-temporal accesses do not displace the
  // lines that ARE re-used (k_f[h] across a head's pairs, the spectrum scratch of the backward) from L2
  static FFC_FN U4 g_r128_nt(const void* base, i32 o16) {
    u32x4v v = __builtin_nontemporal_load(((const u32x4v*)base) + o16);
    return U4{v.x, v.y, v.z, v.w};
  }
  static FFC_FN void g_w128_nt(void* base, i32 o16, U4 v, pred p) {
    if (p) { u32x4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, ((u32x4v*)base) + o16); }
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
  static FFC_FN u32 pk_mul_f16(u32 a, u32 b) {      // v_pk_mul_f16
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) * __builtin_bit_cast(f16x2, b));
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
__device__ __forceinline__ bool map_id(int id, int H, int nchunk, int* h, int* chunk) {
  int xcd = id & 7, s = id >> 3;
  *h = xcd + 8 * (s / nchunk);
  *chunk = s % nchunk;
  return *h < H;
}
// Start-up stagger (tuning flag bits 4..7 of ConvArgs::flags = s): workgroup b waits ((b / 8) mod 8) * s * 512 cycles before its
// first global access.  Every workgroup of a launch runs the same phase sequence for the same time, so without it all CUs
// request their rows in the same instant (a 16 MB burst at the HBM rate = 3.5 us of exposed wait per burst), then all compute.
__device__ __forceinline__ void stagger_start(int flags) {
  const int s = (flags >> 4) & 15;
  if (s == 0) return;
  const int n = ((blockIdx.x >> 3) & 7) * s;
  for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(8);
}
__device__ __forceinline__ bool map_block(int H, int nchunk, int* h, int* chunk) {
  int id = blockIdx.x;
  int xcd = id & 7, s = id >> 3;
  *h = xcd + 8 * (s / nchunk);
  *chunk = s % nchunk;
  return *h < H;
}

// DevB with an opaque lane id: every use site re-derives its lane-dependent values, nothing lane-derived is
// hoisted out of the per-pair loop.  The backward kernels run on a 128-VGPR budget (see Modes::WAcc); hoisted
// invariants there would overflow into the accumulation registers that hold the dk_f partial sums.
struct DevBO : DevB {
  static constexpr bool LEAN_OUTER = true;     // per-tile outer stages (fewer registers)
  // lane id from the execution mask (v_mbcnt: every call site runs with all 64 lanes active) instead of threadIdx.x, so
  // that the work-item id register does not stay live for the whole kernel (under the 128-VGPR budget the allocator
  // parked it in a0 across the pass loop of the multi-pass kernels; build.py check_agpr caught it)
  static FFC_FN i32 lane() {
    int x = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(x));
    return x;
  }
};

// DevBO for launches on 16-byte-aligned tensors with L % 8 == 0 only (the launcher checks: ConvArgs::fast): the multi-pass backward
// kernels (fft 65536 / 131072) -- one copy of the row code per kernel, every row load of a batch in flight (round 5)
struct DevBOF : DevBO {
  static constexpr bool FAST_ONLY = true;
};

}  // namespace ffc

struct ffc_plan {
  ffc::HostPlan hp;
  ffc::HostPlan hp_bf;          // bf16 tables for the dk inverse (fp32 dynamic range), == hp for bf16 plans
  uint8_t* d_blob = nullptr;
  uint8_t* d_blob_bf = nullptr;
  int32_t* d_freq = nullptr;
  int num_cu = 256;
  // tuning knobs, read from the environment ONCE, at plan creation (not on every launch):
  int env_flags = 0;        // FFC_FLAGS    : 2 = k_f streamed, 4 = spectrum scratch streamed (A/B runs)
  int env_stream = -1;      // FFC_STREAM   : 0/1 overrides the streaming (non-temporal) row accesses, -1 = launcher's choice
  int env_persist = -1;     // FFC_PERSIST  : grid cap of the persistent kernels (<= 0: uncapped), -1 = one workgroup per CU
  int env_wg_mult = 0;      // FFC_WG_MULT  : workgroups-per-slot rule of the backward family, 0 = default (2)
};
static inline int ffc_persist(const ffc_plan* p) {
  if (p->env_persist == -1) return p->num_cu & ~7;
  return p->env_persist > 0 ? (p->env_persist & ~7) : 1 << 30;
}

extern "C" void ffc_set_error_(const char* m);
static inline int ffc_fail(const std::string& m) { ffc_set_error_(m.c_str()); return 1; }
#define HIPCHK(x)                                                                     \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) return ffc_fail(std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Dynamic-LDS limit of a kernel: the attribute is per device, so it is set once per (kernel, device) -- a process that drives
// several GPUs (per-device plan cache in flashfftconv.conv.get_plan) would otherwise launch > 64 KB kernels on its second
// device with the first device's setting only.
int ffc_set_lds_once(const void* kernel, int bytes);      // ffc_hip.hip
namespace ffc { struct DkfArgs; }
int ffc_bwdz_launch(int N, int dtype, const ffc::DkfArgs& d, hipStream_t st);      // ffc_k_bwdz.hip: fused backward on saved spectra
template <class K>
static int ffc_set_lds(K kernel, int bytes) { return ffc_set_lds_once((const void*)kernel, bytes); }

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
    FFC_CASE(2048, 1, 32, 32)          // 2 passes of the 1024 kernel (inner-only multi-pass form, Body::InnerPass)
    FFC_CASE(4096, 16, 16, 16)
    FFC_CASE(8192, 32, 16, 16)
    FFC_CASE(16384, 16, 32, 32)
    FFC_CASE(32768, 32, 32, 32)
    FFC_CASE(65536, 32, 32, 32)        // multi-pass sizes: R passes of the 32768 kernel (HostPlan::R, struct Pass)
    FFC_CASE(131072, 32, 32, 32)
  }
#undef FFC_CASE
  return ffc_fail("unsupported fft size");
}

// pairs per chunk / number of chunks so the grid fills the chip (>= ~2 waves of workgroups) while a
// workgroup still loops over several pairs of one head (k_f[h] reuse through L2).
// fp32 dk_f partial-sum slabs per chunk: every geometry reduces its units inside the workgroup (Modes::w_acc_finish,
// Modes::reduce_store_w) and writes one slab.
static inline int ffc_slabs_per_chunk(const ffc_plan* p) { (void)p; return 1; }

static inline void* ffc_zscratch(const ffc_plan* p, void* ws, int H, int nchunk);
static inline void ffc_choose_chunks(const ffc_plan* p, int H, int npair, int* nchunk, int* ppc, bool fwd_only = false) {
  const bool outer = p->hp.N1 > 1;
  int upw = 8 / p->hp.NW;                       // units a workgroup processes per iteration
  int pairs_per_iter = outer ? upw : upw * p->hp.G;
  int wg_per_cu = outer ? 1 : 2;
  int iters_total = (npair + pairs_per_iter - 1) / pairs_per_iter;
  int ipc;
  const int e = p->env_wg_mult;                  // tuning knob of the workgroups-per-slot rule (FFC_WG_MULT)
  if (e > 0 || !fwd_only) {
    // backward family (the kernel, its workspace and dkifft must agree, and every extra chunk is one more fp32 dk_f
    // slab per head): at least `mult` workgroups per slot.  Larger values did not help (profiles/, benchmarks/prof_mult.py).
    int mult = e > 0 ? e : 2;
    int target = p->num_cu * wg_per_cu * mult;
    int nc = (target + H - 1) / H;
    if (nc > iters_total) nc = iters_total;
    if (nc < 1) nc = 1;
    ipc = (iters_total + nc - 1) / nc;
  } else {
    // forward / input-gradient kernel (no per-chunk state): iterations per chunk that minimise
    // (rounds of workgroups over the chip) x (iterations per workgroup + 0.6 for its start-up); ties go to the longer
    // chunks (k_f[h] re-used by one workgroup).  With many heads every choice ties and a workgroup takes all pairs of
    // its head (768 heads x 8 pairs: 3 even rounds); with few heads (a head-sharded rank: 96 or 192 heads x 8 pairs) the
    // pairs are spread so that the last round is not half empty (benchmarks/prof_heads.py, same process: forward 0.169 -> 0.146 ms
    // at B = 16 H = 192, 0.168 -> 0.158 at B = 64 H = 48; unchanged at H = 96 / 384 / 768).
    const long slots = (long)p->num_cu * wg_per_cu;
    long best = -1;
    ipc = iters_total;
    for (int c = 1; c <= iters_total; c++) {
      long nch = (iters_total + c - 1) / c;
      long rounds = ((long)H * nch + slots - 1) / slots;
      long cost = rounds * (10L * c + 6);
      if (best < 0 || cost <= best) { best = cost; ipc = c; }
    }
  }
  *ppc = ipc * pairs_per_iter;
  *nchunk = (npair + *ppc - 1) / *ppc;
}

// spectrum scratch of the backward kernels: behind the fp32 dk_f slabs of the workspace (ffc_dkf_workspace_bytes)
static inline void* ffc_zscratch(const ffc_plan* p, void* ws, int H, int nchunk) {
  return (uint8_t*)ws + (int64_t)nchunk * ffc_slabs_per_chunk(p) * H * p->hp.R * p->hp.NT * 2048 * 4;
}
