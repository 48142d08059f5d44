// CPU wave simulator: runs the very same kernel body (ffc_body.h) with a 64-lane vector backend,
// one host thread per wavefront, pthread barrier = s_barrier, a byte array = LDS.
// TEST INFRASTRUCTURE ONLY (tests/ and __graft_entry__.build use it); the product path is the
// HIP library.  It models the gfx950 primitives the body relies on:
//   v_mfma_f32_32x32x16_{bf16,f16} operand / accumulator lane layouts, ds_read_b64_tr_b16,
//   RNE fp32->bf16/f16 packing, 8-byte predicated global accesses.
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#define FFC_FN inline __attribute__((always_inline))
#include "ffc_body.h"
#include "ffc_modes.h"
#include "ffc_big.h"

namespace ffc {

template <class T>
struct Vec {
  T v[64];
  Vec() {}
  Vec(T s) { for (int i = 0; i < 64; i++) v[i] = s; }
};
#define VOP(op)                                                                           \
  template <class T> inline Vec<T> operator op(const Vec<T>& a, const Vec<T>& b) {        \
    Vec<T> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b.v[i]; return r; }         \
  template <class T> inline Vec<T> operator op(const Vec<T>& a, T b) {                    \
    Vec<T> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] op b; return r; }              \
  template <class T> inline Vec<T> operator op(T a, const Vec<T>& b) {                    \
    Vec<T> r; for (int i = 0; i < 64; i++) r.v[i] = a op b.v[i]; return r; }
VOP(+) VOP(-) VOP(*) VOP(/) VOP(%) VOP(&) VOP(|) VOP(^)
#undef VOP
template <class T> inline Vec<T> operator<<(const Vec<T>& a, int s) { Vec<T> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] << s; return r; }
template <class T> inline Vec<T> operator>>(const Vec<T>& a, int s) { Vec<T> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] >> s; return r; }
// mixed u32 vector with int literal masks
inline Vec<uint32_t> operator&(const Vec<uint32_t>& a, unsigned b) { Vec<uint32_t> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] & b; return r; }
inline Vec<int> operator*(const Vec<int>& a, long b) { return a * (int)b; }
inline Vec<bool> operator<(const Vec<int>& a, int b) { Vec<bool> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] < b; return r; }
inline Vec<bool> operator<(const Vec<int>& a, const Vec<int>& b) { Vec<bool> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] < b.v[i]; return r; }
inline Vec<bool> operator>=(const Vec<int>& a, int b) { Vec<bool> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] >= b; return r; }
inline Vec<bool> operator&&(const Vec<bool>& a, const Vec<bool>& b) { Vec<bool> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
inline Vec<bool> operator&&(const Vec<bool>& a, bool b) { Vec<bool> r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] && b; return r; }

struct WgCtx {
  std::vector<uint8_t> lds;
  pthread_barrier_t bar;
  int nwaves;
};
static thread_local WgCtx* g_wg = nullptr;
static thread_local int g_wave = 0;
static thread_local Vec<float> g_agpr[128];   // per-wave accumulation registers (DevB: a0..a127)
static std::atomic<long> g_dma_count{0};      // LDS-DMA instructions executed (tests: the path under test really ran)

static inline float dt_to_f32(int DT, uint16_t h) { return DT == DT_BF16 ? bf16_to_f32(h) : f16_to_f32(h); }
static inline uint16_t f32_to_dt(int DT, float f) { return DT == DT_BF16 ? f32_to_bf16(f) : f32_to_f16(f); }

struct SimB {
  static constexpr bool LEAN_OUTER = false;
  static constexpr bool FAST_ONLY = false;
  using f32 = Vec<float>;
  using i32 = Vec<int>;
  using u32 = Vec<uint32_t>;
  using pred = Vec<bool>;
  struct U2 { u32 x, y; };
  struct U4 { u32 x, y, z, w; };
  struct A16 { f32 v[16]; f32& operator[](int i) { return v[i]; } const f32& operator[](int i) const { return v[i]; } };
  struct W4 { u32 v[4]; u32& operator[](int i) { return v[i]; } const u32& operator[](int i) const { return v[i]; } };
  template <bool CONJ> static void cmul16(A16& re, A16& im, const A16& tr, const A16& ti) {
    for (int r = 0; r < 16; r++) {
      f32 a = re[r], b = im[r];
      if (!CONJ) { re[r] = a * tr[r] - b * ti[r]; im[r] = a * ti[r] + b * tr[r]; }
      else { re[r] = a * tr[r] + b * ti[r]; im[r] = b * tr[r] - a * ti[r]; }
    }
  }
  struct F2 { f32 x, y; };
  static F2 f2(const f32& a, const f32& b) { F2 v; v.x = a; v.y = b; return v; }
  static f32 f2_lo(const F2& v) { return v.x; }
  static f32 f2_hi(const F2& v) { return v.y; }
  static void cmulp(const F2& xr, const F2& xi, const f32& wr, const f32& wi, F2& yr, F2& yi) {
    F2 r, i;
    r.x = xr.x * wr - xi.x * wi; r.y = xr.y * wr - xi.y * wi;
    i.x = xr.x * wi + xi.x * wr; i.y = xr.y * wi + xi.y * wr;
    yr = r; yi = i;
  }
  template <bool CONJ>
  static void cmul2v(A16& re, A16& im, int r0, const F2& tr, const F2& ti) { cmul2<CONJ>(re, im, r0, tr.x, tr.y, ti.x, ti.y); }
  static void cmac2_conj(F2& wr, F2& wi, const A16& a, const A16& b, int r0, const F2& zr, const F2& zi) {
    wr.x = wr.x + (a[r0] * zr.x + b[r0] * zi.x);
    wr.y = wr.y + (a[r0 + 1] * zr.y + b[r0 + 1] * zi.y);
    wi.x = wi.x + (b[r0] * zr.x - a[r0] * zi.x);
    wi.y = wi.y + (b[r0 + 1] * zr.y - a[r0 + 1] * zi.y);
  }
  static void agpr_reserve() {}
  template <int I> static f32 agpr_get() { return g_agpr[I]; }
  template <int I> static void agpr_set(const f32& x) { g_agpr[I] = x; }
  // DevB::mfma_acc_bf16: a[I0 .. I0+15] += A x B (bf16 operands); defined behind mfma<>
  template <int I0> static void mfma_acc_bf16(const struct W4& a, const struct W4& b);
  static void mfma_settle() {}
  static void pin(W4&) {}
  static void pin4(U4&) {}
  static void sched_fence() {}
  template <bool CONJ>
  static void cmul2(A16& re, A16& im, int r0, const f32& tr0, const f32& tr1, const f32& ti0, const f32& ti1) {
    const f32* tr[2] = {&tr0, &tr1}; const f32* ti[2] = {&ti0, &ti1};
    for (int q = 0; q < 2; q++) {
      f32 a = re[r0 + q], b = im[r0 + q];
      if (!CONJ) { re[r0 + q] = a * *tr[q] - b * *ti[q]; im[r0 + q] = a * *ti[q] + b * *tr[q]; }
      else { re[r0 + q] = a * *tr[q] + b * *ti[q]; im[r0 + q] = b * *tr[q] - a * *ti[q]; }
    }
  }
  static A16 a16_scale(const A16& a, float s) { A16 r; for (int i = 0; i < 16; i++) r[i] = a[i] * f32(s); return r; }
  static A16 a16_zero() { A16 z; for (int i = 0; i < 16; i++) z.v[i] = f32(0.f); return z; }
  static W4 w4(const u32& a, const u32& b, const u32& c, const u32& e) { W4 v; v.v[0] = a; v.v[1] = b; v.v[2] = c; v.v[3] = e; return v; }
  static bool HAS_TR;

  static i32 lane() { i32 r; for (int i = 0; i < 64; i++) r.v[i] = i; return r; }
  static int wave() { return g_wave; }
  static void barrier() { if (g_wg->nwaves > 1) pthread_barrier_wait(&g_wg->bar); }
  static f32 fconst(float c) { return f32(c); }
  static pred ptrue() { return pred(true); }
  static pred pfalse() { return pred(false); }
  static f32 as_f32(const u32& a) { f32 r; for (int i = 0; i < 64; i++) memcpy(&r.v[i], &a.v[i], 4); return r; }
  static u32 as_u32(const f32& a) { u32 r; for (int i = 0; i < 64; i++) memcpy(&r.v[i], &a.v[i], 4); return r; }

  static uint8_t* L() { return g_wg->lds.data(); }
  static void chk(int off, int n) { if (off < 0 || off + n > (int)g_wg->lds.size() || (off & (n - 1))) abort(); }
  static U2 lds_r64(const i32& off) {
    U2 r;
    for (int i = 0; i < 64; i++) { chk(off.v[i], 8); memcpy(&r.x.v[i], L() + off.v[i], 4); memcpy(&r.y.v[i], L() + off.v[i] + 4, 4); }
    return r;
  }
  static void lds_w64(const i32& off, const U2& v) {
    for (int i = 0; i < 64; i++) { chk(off.v[i], 8); memcpy(L() + off.v[i], &v.x.v[i], 4); memcpy(L() + off.v[i] + 4, &v.y.v[i], 4); }
  }
  static void lds_w32(const i32& off, const u32& v) {
    for (int i = 0; i < 64; i++) { chk(off.v[i], 4); memcpy(L() + off.v[i], &v.v[i], 4); }
  }
  static void lds_w128(const i32& off, const U4& v, const pred& p) {
    for (int i = 0; i < 64; i++)
      if (p.v[i]) {
        chk(off.v[i], 16);
        memcpy(L() + off.v[i], &v.x.v[i], 4); memcpy(L() + off.v[i] + 4, &v.y.v[i], 4);
        memcpy(L() + off.v[i] + 8, &v.z.v[i], 4); memcpy(L() + off.v[i] + 12, &v.w.v[i], 4);
      }
  }
  static U4 lds_r128(const i32& off) {
    U4 r;
    for (int i = 0; i < 64; i++) {
      chk(off.v[i], 16);
      memcpy(&r.x.v[i], L() + off.v[i], 4); memcpy(&r.y.v[i], L() + off.v[i] + 4, 4);
      memcpy(&r.z.v[i], L() + off.v[i] + 8, 4); memcpy(&r.w.v[i], L() + off.v[i] + 12, 4);
    }
    return r;
  }
  static void lds_w16(const i32& off, const u32& v) {
    for (int i = 0; i < 64; i++) { chk(off.v[i], 2); uint16_t h = (uint16_t)v.v[i]; memcpy(L() + off.v[i], &h, 2); }
  }
  static u32 lds_r32(const i32& off) {
    u32 r;
    for (int i = 0; i < 64; i++) { chk(off.v[i], 4); memcpy(&r.v[i], L() + off.v[i], 4); }
    return r;
  }
  static u32 lds_r16(const i32& off) {
    u32 r;
    for (int i = 0; i < 64; i++) { chk(off.v[i], 2); uint16_t h; memcpy(&h, L() + off.v[i], 2); r.v[i] = h; }
    return r;
  }
  // ds_read_b64_tr_b16: within each 16-lane group lane i' supplies the address of 4 b16 values
  // M[i'][0..3]; lane i receives elements j=0..3 = M[4j + (i>>2)][i&3].
  static U2 lds_r64_tr(const i32& off) {
    U2 r;
    for (int g = 0; g < 4; g++) {
      uint16_t M[16][4];
      for (int i = 0; i < 16; i++) { chk(off.v[g * 16 + i], 8); memcpy(M[i], L() + off.v[g * 16 + i], 8); }
      for (int i = 0; i < 16; i++) {
        uint16_t e[4];
        for (int j = 0; j < 4; j++) e[j] = M[4 * j + (i >> 2)][i & 3];
        r.x.v[g * 16 + i] = e[0] | ((uint32_t)e[1] << 16);
        r.y.v[g * 16 + i] = e[2] | ((uint32_t)e[3] << 16);
      }
    }
    return r;
  }
  static void lds_fence() {}
  // LDS-DMA model: executed at issue (the simulator has no asynchronous memory pipe; the ordering rules -- vmcnt before the
  // wave's own reads -- are the device code's business)
  template <bool NT>
  static void g2lds32(const void* base, const i32& dw, int lds_off) {
    g_dma_count++;
    for (int i = 0; i < 64; i++) { chk(lds_off + 4 * i, 4); memcpy(L() + lds_off + 4 * i, (const uint32_t*)base + dw.v[i], 4); }
  }
  template <bool NT>
  static void g2lds128(const void* base, const i32& o16, int lds_off) {
    g_dma_count++;
    for (int i = 0; i < 64; i++) { chk(lds_off + 16 * i, 16); memcpy(L() + lds_off + 16 * i, (const uint8_t*)base + 16 * (int64_t)o16.v[i], 16); }
  }
  static void vm_wait0() {}
  static void lds_w32p(const i32& off, const u32& v, const pred& p) {
    for (int i = 0; i < 64; i++) if (p.v[i]) { chk(off.v[i], 4); memcpy(L() + off.v[i], &v.v[i], 4); }
  }
  static pred pnot(const pred& p) { pred r; for (int i = 0; i < 64; i++) r.v[i] = !p.v[i]; return r; }
  static i32 mul24(const i32& a, const i32& b) { return a * b; }
  static unsigned long long clock() { return 0; }
  template <int P> static void setprio() {}
  static U2 u2_from64(unsigned long long v) { U2 r; r.x = u32((uint32_t)v); r.y = u32((uint32_t)(v >> 32)); return r; }
  static void settle(f32&, f32&) {}
  static f32 i2f(const i32& a) { f32 r; for (int i = 0; i < 64; i++) r.v[i] = (float)a.v[i]; return r; }
  static f32 cos_rev(const f32& a) { f32 r; for (int i = 0; i < 64; i++) r.v[i] = (float)cos(6.283185307179586 * (double)a.v[i]); return r; }
  static f32 sin_rev(const f32& a) { f32 r; for (int i = 0; i < 64; i++) r.v[i] = (float)sin(6.283185307179586 * (double)a.v[i]); return r; }
  static i32 opaque(const i32& x) { return x; }
  static u32 uconst(uint32_t c) { return u32(c); }
  static i32 imin(const i32& a, int b) { i32 r; for (int i = 0; i < 64; i++) r.v[i] = a.v[i] < b ? a.v[i] : b; return r; }
  static u32 merge_lo(const u32& a, const u32& b) { u32 r; for (int i = 0; i < 64; i++) r.v[i] = (a.v[i] & 0xffffu) | (b.v[i] << 16); return r; }
  static u32 merge_hi(const u32& a, const u32& b) { u32 r; for (int i = 0; i < 64; i++) r.v[i] = (a.v[i] >> 16) | (b.v[i] & 0xffff0000u); return r; }
  static u32 sel(const pred& p, const u32& a, const u32& b) { u32 r; for (int i = 0; i < 64; i++) r.v[i] = p.v[i] ? a.v[i] : b.v[i]; return r; }
  static u32 g_r16(const void* base, const i32& e, const pred& p) {
    u32 r;
    for (int i = 0; i < 64; i++) { uint16_t h = 0; if (p.v[i]) memcpy(&h, (const uint8_t*)base + (int64_t)e.v[i] * 2, 2); r.v[i] = h; }
    return r;
  }
  static void g_w16(void* base, const i32& e, const u32& v, const pred& p) {
    for (int i = 0; i < 64; i++) if (p.v[i]) { uint16_t h = (uint16_t)v.v[i]; memcpy((uint8_t*)base + (int64_t)e.v[i] * 2, &h, 2); }
  }
  static u32 g_r32(const void* base, const i32& e, const pred& p) {
    u32 r;
    for (int i = 0; i < 64; i++) { uint32_t h = 0; if (p.v[i]) memcpy(&h, (const uint8_t*)base + (int64_t)e.v[i] * 4, 4); r.v[i] = h; }
    return r;
  }
  static void g_w32(void* base, const i32& e, const u32& v, const pred& p) {
    for (int i = 0; i < 64; i++) if (p.v[i]) memcpy((uint8_t*)base + (int64_t)e.v[i] * 4, &v.v[i], 4);
  }
  static U2 g_r64(const void* base, const i32& o8, const pred& p) {
    U2 r;
    for (int i = 0; i < 64; i++) {
      if (p.v[i]) { const uint8_t* q = (const uint8_t*)base + (int64_t)o8.v[i] * 8; memcpy(&r.x.v[i], q, 4); memcpy(&r.y.v[i], q + 4, 4); }
      else { r.x.v[i] = 0; r.y.v[i] = 0; }
    }
    return r;
  }
  static void g_w64(void* base, const i32& o8, const U2& v, const pred& p) {
    for (int i = 0; i < 64; i++)
      if (p.v[i]) { uint8_t* q = (uint8_t*)base + (int64_t)o8.v[i] * 8; memcpy(q, &v.x.v[i], 4); memcpy(q + 4, &v.y.v[i], 4); }
  }
  static U4 g_r128(const void* base, const i32& o16) {
    U4 r;
    for (int i = 0; i < 64; i++) {
      const uint8_t* q = (const uint8_t*)base + (int64_t)o16.v[i] * 16;
      memcpy(&r.x.v[i], q, 4); memcpy(&r.y.v[i], q + 4, 4); memcpy(&r.z.v[i], q + 8, 4); memcpy(&r.w.v[i], q + 12, 4);
    }
    return r;
  }
  static U4 g_r128_nt(const void* base, const i32& o16) { return g_r128(base, o16); }
  static U4 g_r128p(const void* base, const i32& o16, const pred& p) {
    U4 r;
    for (int i = 0; i < 64; i++) {
      if (p.v[i]) {
        const uint8_t* q = (const uint8_t*)base + (int64_t)o16.v[i] * 16;
        memcpy(&r.x.v[i], q, 4); memcpy(&r.y.v[i], q + 4, 4); memcpy(&r.z.v[i], q + 8, 4); memcpy(&r.w.v[i], q + 12, 4);
      } else { r.x.v[i] = r.y.v[i] = r.z.v[i] = r.w.v[i] = 0; }
    }
    return r;
  }
  static void g_w128(void* base, const i32& o16, const U4& v, const pred& p) {
    for (int i = 0; i < 64; i++)
      if (p.v[i]) {
        uint8_t* q = (uint8_t*)base + (int64_t)o16.v[i] * 16;
        memcpy(q, &v.x.v[i], 4); memcpy(q + 4, &v.y.v[i], 4); memcpy(q + 8, &v.z.v[i], 4); memcpy(q + 12, &v.w.v[i], 4);
      }
  }
  static void g_w128_nt(void* base, const i32& o16, const U4& v, const pred& p) { g_w128(base, o16, v, p); }
  // D[i][j] += sum_k A[i][k] B[k][j];  A lane l: A[l&31][8*(l>>5)+e];  B lane l: B[8*(l>>5)+e][l&31];
  // D lane l reg r: D[acc_row(r, l>>5)][l&31].
  template <int DT>
  static void mfma(A16& acc, const W4& a, const W4& b) {
    float A[32][16], Bm[16][32];
    for (int l = 0; l < 64; l++)
      for (int e = 0; e < 8; e++) {
        uint16_t ha = (uint16_t)(a[e >> 1].v[l] >> (16 * (e & 1)));
        uint16_t hb = (uint16_t)(b[e >> 1].v[l] >> (16 * (e & 1)));
        A[l & 31][8 * (l >> 5) + e] = dt_to_f32(DT, ha);
        Bm[8 * (l >> 5) + e][l & 31] = dt_to_f32(DT, hb);
      }
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) {
        int i = acc_row(r, l >> 5), j = l & 31;
        float s = 0;
        for (int k = 0; k < 16; k++) s += A[i][k] * Bm[k][j];
        acc[r].v[l] += s;
      }
  }
  template <int DT> static u32 pack(const f32& lo, const f32& hi) {
    u32 r;
    for (int i = 0; i < 64; i++) r.v[i] = f32_to_dt(DT, lo.v[i]) | ((uint32_t)f32_to_dt(DT, hi.v[i]) << 16);
    return r;
  }
  // v_pk_mul_f16: the exact fp32 product of two fp16 values rounded once to fp16 (= the native fp16 multiply)
  static u32 pk_mul_f16(const u32& a, const u32& b) {
    u32 r;
    for (int i = 0; i < 64; i++) {
      float lo = f16_to_f32((uint16_t)a.v[i]) * f16_to_f32((uint16_t)b.v[i]);
      float hi = f16_to_f32((uint16_t)(a.v[i] >> 16)) * f16_to_f32((uint16_t)(b.v[i] >> 16));
      r.v[i] = f32_to_f16(lo) | ((uint32_t)f32_to_f16(hi) << 16);
    }
    return r;
  }
  template <int DT> static f32 unpack_lo(const u32& a) { f32 r; for (int i = 0; i < 64; i++) r.v[i] = dt_to_f32(DT, (uint16_t)a.v[i]); return r; }
  template <int DT> static f32 unpack_hi(const u32& a) { f32 r; for (int i = 0; i < 64; i++) r.v[i] = dt_to_f32(DT, (uint16_t)(a.v[i] >> 16)); return r; }
};
template <int I0> void SimB::mfma_acc_bf16(const SimB::W4& a, const SimB::W4& b) {
  A16 acc;
  for (int r = 0; r < 16; r++) acc[r] = g_agpr[I0 + r];
  mfma<DT_BF16>(acc, a, b);
  for (int r = 0; r < 16; r++) g_agpr[I0 + r] = acc[r];
}
bool SimB::HAS_TR = true;
// mirrors DevBO: the backward kernels use the per-tile outer stages
struct SimBO : SimB { static constexpr bool LEAN_OUTER = true; };
struct SimBOF : SimBO { static constexpr bool FAST_ONLY = true; };      // the fast-only instantiation of the multi-pass backward (DevBOF)
static bool g_force_slow = false;

// Run `fn(wg_index)` for one workgroup of nwaves wavefronts with lds_bytes of LDS.
template <class F>
static void run_wg(int nwaves, int lds_bytes, F fn) {
  WgCtx ctx;
  ctx.lds.assign(lds_bytes, 0xCD);
  ctx.nwaves = nwaves;
  pthread_barrier_init(&ctx.bar, nullptr, nwaves);
  std::vector<std::thread> th;
  for (int w = 0; w < nwaves; w++)
    th.emplace_back([&, w]() { g_wg = &ctx; g_wave = w; fn(); });
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&ctx.bar);
}

template <class GEO, int DT>
static void sim_conv_t(const ConvArgs& a) {
  for (int h = 0; h < a.H; h++)
    for (int c = 0; c < a.nchunk; c++) {
      if constexpr (GEO::N == 32768) {
        if (a.R > 1) {       // mirrors conv_kernel<..., RP = true>: the passes run one after the other in the same workgroup
          run_wg(GEO::WGW, GEO::LDS_BYTES, [&]() {
            Body<SimB, GEO, DT>::setup_tables(a.tab, a.t);
            if (16 * GEO::Mi >= a.L) Body<SimB, GEO, DT>::template conv_job<true, true>(a, h, c);
            else Body<SimB, GEO, DT>::template conv_job<false, true>(a, h, c);
          });
          continue;
        }
      }
      if constexpr (GEO::N == 1024) {
        if (a.R > 1) {       // inner-only multi-pass form (fft 2048)
          run_wg(GEO::WGW, GEO::LDS_BYTES + a.R * Body<SimB, GEO, DT>::IPASS_BYTES, [&]() {
            Body<SimB, GEO, DT>::setup_tables(a.tab, a.t);
            Body<SimB, GEO, DT>::setup_tables_ipass(a.tab, a.t, a.R);
            if (a.zsave || a.yraw) Body<SimB, GEO, DT>::template conv_job<false, true, true>(a, h, c);      // conv_rp_kernel<.., SZ>
            else Body<SimB, GEO, DT>::template conv_job<false, true>(a, h, c);
          });
          continue;
        }
      }
      run_wg(GEO::WGW, GEO::LDS_BYTES, [&]() {
        if constexpr (GEO::HAS_SP) {      // frequency-sparse variant (ffc_conv_fwd_sparse)
          if (a.sparse) {
            if ((GEO::N1 / 2) * GEO::Mi >= a.L) Body<SimB, GEO, DT>::template conv<true, false, true>(a, h, c);
            else Body<SimB, GEO, DT>::template conv<false, false, true>(a, h, c);
            return;
          }
        }
        if constexpr (GEO::OUTER && GEO::NW > 1) {
          if (a.kfuse_k || a.kfuse_x) {   // k -> k_f of this head inside the same workgroup (conv_kernel, ConvArgs::kfuse_k / kfuse_x)
            Body<SimB, GEO, DT>::setup_tables(a.tab, a.t);
            Modes<SimB, GEO, DT>::kfft_head(a, h);
            const bool half = (GEO::N1 / 2) * GEO::Mi >= a.L;
            if (a.zsave) { if (half) Body<SimB, GEO, DT>::template conv_job<true, false, true>(a, h, c); else Body<SimB, GEO, DT>::template conv_job<false, false, true>(a, h, c); }
            else { if (half) Body<SimB, GEO, DT>::template conv_job<true>(a, h, c); else Body<SimB, GEO, DT>::template conv_job<false>(a, h, c); }
            return;
          }
        }
        if constexpr (GEO::OUTER) {       // the launcher's HALF variant
          if (a.zsave) {                  // spectrum-saving training forward (ffc_conv_fwd_z)
            if ((GEO::N1 / 2) * GEO::Mi >= a.L) Body<SimB, GEO, DT>::template conv<true, true>(a, h, c);
            else Body<SimB, GEO, DT>::template conv<false, true>(a, h, c);
            return;
          }
          if ((GEO::N1 / 2) * GEO::Mi >= a.L) { Body<SimB, GEO, DT>::template conv<true>(a, h, c); return; }
        } else {
          // single-tile sizes: the training forward that keeps the spectra and / or the output before the postgate (conv_kernel<.., SZ>)
          if (a.zsave || a.yraw) { Body<SimB, GEO, DT>::template conv<false, true>(a, h, c); return; }
        }
        Body<SimB, GEO, DT>::conv(a, h, c);
      });
    }
}

template <template <class, int> class FN, class... A>
static int dispatch(int N, int dtype, A&&... args) {
#define FFC_CASE(NN, a, b, c)                                                     \
  case NN:                                                                        \
    if (dtype == DT_BF16) FN<Geo<a, b, c>, DT_BF16>::run(args...);                \
    else FN<Geo<a, b, c>, DT_F16>::run(args...);                                  \
    return 0;
  switch (N) {
    FFC_CASE(256, 1, 16, 16)
    FFC_CASE(512, 1, 16, 32)
    FFC_CASE(1024, 1, 32, 32)
    FFC_CASE(2048, 1, 32, 32)
    FFC_CASE(4096, 16, 16, 16)
    FFC_CASE(8192, 32, 16, 16)
    FFC_CASE(16384, 16, 32, 32)
    FFC_CASE(32768, 32, 32, 32)
    FFC_CASE(65536, 32, 32, 32)       // multi-pass sizes: R passes of the 32768 kernel (HostPlan::R, struct Pass)
    FFC_CASE(131072, 32, 32, 32)
  }
#undef FFC_CASE
  return -1;
}
template <class GEO, int DT> struct ConvRun { static void run(const ConvArgs& a) { sim_conv_t<GEO, DT>(a); } };
template <class GEO, int DT> struct KfRun {
  static void run(const KfArgs& a) {
    const int nunits = GEO::OUTER ? a.H : (a.H + GEO::G - 1) / GEO::G;
    for (int wg = 0; wg < (nunits + GEO::UPW - 1) / GEO::UPW; wg++)
      run_wg(GEO::WGW, GEO::LDS_BYTES + (GEO::OUTER ? 0 : 4 * Body<SimB, GEO, DT>::IPASS_BYTES), [&]() { Modes<SimB, GEO, DT>::kfft(a, wg); });
  }
};
template <class GEO, int DT> struct DkfRun {
  static void run(const DkfArgs& d) {
    for (int h = 0; h < d.c.H; h++)
      for (int c = 0; c < d.c.nchunk; c++)
        run_wg(GEO::WGW, GEO::LDS_BYTES + (GEO::OUTER ? 0 : 4 * Body<SimB, GEO, DT>::IPASS_BYTES), [&]() {
          if constexpr (GEO::N == 32768) {
            if (d.c.R > 1) {
              Modes<SimBO, GEO, DT>::BD::setup_tables(d.c.tab, d.c.t);
              const bool half = 16 * GEO::Mi >= d.c.L;
              for (int k0 = 0; k0 < d.c.R; k0++) {
                if (half) Modes<SimBO, GEO, DT>::template dkf<true, true>(d, h, c, h * d.c.nchunk + c, k0, SimBO::wave());
                else Modes<SimBO, GEO, DT>::template dkf<false, true>(d, h, c, h * d.c.nchunk + c, k0, SimBO::wave());
              }
              return;
            }
          }
          if constexpr (GEO::OUTER) {     // the launcher's HALF variant (L <= N/2)
            if ((GEO::N1 / 2) * GEO::Mi >= d.c.L) { Modes<SimBO, GEO, DT>::template dkf<true>(d, h, c, h * d.c.nchunk + c); return; }
          }
          Modes<SimBO, GEO, DT>::dkf(d, h, c, h * d.c.nchunk + c);
        });
  }
};
template <class GEO, int DT> struct BwdRun {
  static void run(const DkfArgs& d) {
    for (int h = 0; h < d.c.H; h++)
      for (int c = 0; c < d.c.nchunk; c++)
        run_wg(GEO::WGW, GEO::LDS_BYTES + (GEO::OUTER ? 0 : 4 * Body<SimB, GEO, DT>::IPASS_BYTES), [&]() {
          if constexpr (GEO::N == 32768) {
            if (d.c.R > 1) {
              Modes<SimBO, GEO, DT>::BD::setup_tables(d.c.tab, d.c.t);
              const bool half = 16 * GEO::Mi >= d.c.L;
              for (int k0 = 0; k0 < d.c.R; k0++) {
                if (d.c.fast && (FFC_RP_FASTK != 0)) {      // the launcher's fast-only kernel (bwd_rp_kernel<.., FASTK = true>)
                  if (half) Modes<SimBOF, GEO, DT>::template bwd<true, true>(d, h, c, h * d.c.nchunk + c, k0, SimBO::wave());
                  else Modes<SimBOF, GEO, DT>::template bwd<false, true>(d, h, c, h * d.c.nchunk + c, k0, SimBO::wave());
                } else if (half) Modes<SimBO, GEO, DT>::template bwd<true, true>(d, h, c, h * d.c.nchunk + c, k0, SimBO::wave());
                else Modes<SimBO, GEO, DT>::template bwd<false, true>(d, h, c, h * d.c.nchunk + c, k0, SimBO::wave());
              }
              return;
            }
          }
          if constexpr (GEO::OUTER) {
            // (the library's launcher runs the saved-spectra form as its own instantiation, ZM = 1: ffc_k_bwdz.hip)
            if (d.zin) {
              if ((GEO::N1 / 2) * GEO::Mi >= d.c.L) Modes<SimBO, GEO, DT>::template bwd<true, false, true, 1>(d, h, c, h * d.c.nchunk + c);
              else Modes<SimBO, GEO, DT>::template bwd<false, false, true, 1>(d, h, c, h * d.c.nchunk + c);
              return;
            }
            if ((GEO::N1 / 2) * GEO::Mi >= d.c.L) { Modes<SimBO, GEO, DT>::template bwd<true>(d, h, c, h * d.c.nchunk + c); return; }
          }
          Modes<SimBO, GEO, DT>::bwd(d, h, c, h * d.c.nchunk + c);
        });
  }
};
template <class GEO, int DT> struct DkRun {
  static void run(const DkArgs& a) {
    const int nunits = GEO::OUTER ? a.H : (a.H + GEO::G - 1) / GEO::G;
    for (int wg = 0; wg < (nunits + GEO::UPW - 1) / GEO::UPW; wg++)
      run_wg(GEO::WGW, GEO::LDS_BYTES + (GEO::OUTER ? 0 : 4 * Body<SimB, GEO, DT>::IPASS_BYTES), [&]() { Modes<SimB, GEO, DT>::dkifft(a, wg); });
  }
};
template <class GEO, int DT> struct UpwGet { static void run(int* out) { *out = GEO::UPW; } };

}  // namespace ffc

using namespace ffc;

extern "C" {

void ffcsim_set_tr(int on) { SimB::HAS_TR = on != 0; }
void ffcsim_force_slow_io(int on) { g_force_slow = on != 0; }

// Mirror of ffc_selftest_primitives (ffc_hip.hip selftest_kernel) on the simulator backend.
int ffcsim_selftest_primitives(const uint32_t* in, uint32_t* out) {
  run_wg(1, 1024, [&]() {
    using Bk = SimB;
    Bk::W4 a, b;
    for (int i = 0; i < 4; i++) for (int l = 0; l < 64; l++) { a[i].v[l] = in[l * 8 + i]; b[i].v[l] = in[l * 8 + 4 + i]; }
    Bk::A16 acc = Bk::a16_zero(), acc2 = Bk::a16_zero();
    Bk::mfma<DT_BF16>(acc, a, b);
    Bk::mfma<DT_F16>(acc2, a, b);
    Bk::i32 lane = Bk::lane();
    Bk::U2 w; w.x = a[0]; w.y = a[1];
    Bk::lds_w64(lane * 8, w);
    Bk::U2 t = Bk::lds_r64_tr(((lane * 5 + 3) & 63) * 8);
    Bk::u32 p0 = Bk::pack<DT_BF16>(acc[0], acc[1]), p1 = Bk::pack<DT_F16>(acc[0], acc[1]);
    Bk::f32 u0 = Bk::unpack_lo<DT_F16>(a[2]), u1 = Bk::unpack_hi<DT_F16>(a[2]);
    Bk::f32 u2 = Bk::unpack_lo<DT_BF16>(a[2]), u3 = Bk::unpack_hi<DT_BF16>(a[2]);
    for (int l = 0; l < 64; l++) {
      for (int i = 0; i < 16; i++) { memcpy(&out[l * 40 + i], &acc[i].v[l], 4); memcpy(&out[l * 40 + 16 + i], &acc2[i].v[l], 4); }
      out[l * 40 + 32] = t.x.v[l]; out[l * 40 + 33] = t.y.v[l];
      out[l * 40 + 34] = p0.v[l]; out[l * 40 + 35] = p1.v[l];
      memcpy(&out[l * 40 + 36], &u0.v[l], 4); memcpy(&out[l * 40 + 37], &u1.v[l], 4);
      memcpy(&out[l * 40 + 38], &u2.v[l], 4); memcpy(&out[l * 40 + 39], &u3.v[l], 4);
    }
  });
  return 0;
}

// Plan introspection (also used by tests to build k_f in internal order on the host).
int ffcsim_plan_info(int N, int dtype, int* nt, double* s_fwd, double* s_k, int32_t* kf_freq /* nt*1024 or null */) {
  HostPlan p;
  if (!build_plan(N, dtype, &p)) return -1;
  *nt = p.NT * p.R; *s_fwd = p.s_fwd; *s_k = p.s_k;      // k_f tiles per head (R passes x NT)
  if (kf_freq) memcpy(kf_freq, p.kf_freq.data(), p.kf_freq.size() * 4);
  return 0;
}

// Same contract as ffc_conv_fwd (include/flashfftconv_hip.h) but on host memory.
static int g_sparse_rows = 0;
void ffcsim_set_sparse(int rows) { g_sparse_rows = rows; }      // next ffcsim_conv_fwd calls run the frequency-sparse variant
// spectrum buffer ([H][npair][N] complex dtype pairs) / pre-postgate output of the NEXT ffcsim_conv_fwd (written) and
// ffcsim_conv_bwd (read) calls: the ffc_conv_fwd_z / ffc_conv_bwd_z(y) pair.  Fused single-pass sizes >= 4096.  flags: ConvArgs::flags.
static void* g_z = nullptr; static void* g_yraw = nullptr; static int g_flags = 0;
void ffcsim_set_z(void* z, void* yraw, int flags) { g_z = z; g_yraw = yraw; g_flags = flags; }
long ffcsim_dma_count() { return g_dma_count.exchange(0); }
// dk (H, Lk) fp32 written by the NEXT ffcsim_conv_bwd itself (DkfArgs::dk_out, Modes::dk_tail; the caller passes nchunk = 1)
static float* g_dk_out = nullptr; static int g_dk_lk = 0;
void ffcsim_set_fused_dk(float* dk, int Lk) { g_dk_out = dk; g_dk_lk = Lk; }
// k (H, Lk) fp32 transformed by the NEXT ffcsim_conv_fwd itself into its kf argument (ConvArgs::kfuse_k, Modes::kfft_head)
static const float* g_kfuse_k = nullptr; static int g_kfuse_lk = 0;
void ffcsim_set_fused_k(const float* k, int Lk) { g_kfuse_k = k; g_kfuse_lk = Lk; }
// complex forms (HBM-level sizes): k_f rows from a pair-plane tensor (ConvArgs::kfuse_x), dk rows into one (DkfArgs::dk_pair)
static const void* g_kfuse_x = nullptr; static float g_kfuse_xscale = 1.f;
void ffcsim_set_fused_kx(const void* xpair, float scale) { g_kfuse_x = xpair; g_kfuse_xscale = scale; }
static void* g_dk_pair = nullptr; static float g_dk_pair_scale = 1.f;
void ffcsim_set_fused_dkpair(void* outpair, float scale) { g_dk_pair = outpair; g_dk_pair_scale = scale; }
static int g_big_pipe = 0;      // as in the library: opt-in
void ffcsim_set_big_pipe(int on) { g_big_pipe = on; }      // 0: every outer pass through BigBody::run (one block per workgroup)
int ffcsim_conv_fwd(int N, int dtype, const void* u, const void* kf, const void* pregate, const void* postgate,
                    void* y, int B, int H, int L, int conj_kf) {
  HostPlan p;
  if (!build_plan(N, dtype, &p)) return -1;
  if (L > N || L <= 0) return -2;
  ConvArgs a{};
  a.u = u; a.pregate = pregate; a.postgate = postgate; a.y = y; a.kf = kf;
  a.tab = p.blob.data(); a.t = p.tabs;
  a.B = B; a.H = H; a.L = L; a.npair = (B + 1) / 2;
  a.sbu = a.sbg = a.sbp = a.sby = (int64_t)H * L;
  a.nchunk = 1; a.ppc = a.npair; a.conj_kf = conj_kf; a.s_inv = (float)p.s_inv; a.s_fwd = (float)p.s_fwd;
  a.fast = (L % 8 == 0) && !g_force_slow;
  a.R = p.R;
  a.sparse = g_sparse_rows;
  if (p.N1 > 1 && p.R == 1) { a.zsave = g_z; a.yraw = g_z ? g_yraw : nullptr; }
  if (p.N1 <= 1) { a.zsave = g_z; a.yraw = g_yraw; }      // single-tile sizes: either one alone too (conv_fwd_impl)
  if (g_kfuse_x && N >= 8192 && N <= 32768 && !a.sparse) {
    a.kfuse_x = g_kfuse_x; a.kfuse_scale = g_kfuse_xscale; a.kfuse_Lk = N; a.kfuse_fast = 1;
  }
  if (g_kfuse_k && N >= 8192 && N <= 32768 && !a.sparse) {
    a.kfuse_k = g_kfuse_k; a.kfuse_Lk = g_kfuse_lk; a.kfuse_scale = (float)(p.s_k / p.s_fwd) / (dtype == DT_F16 ? 256.f : 1.f); a.kfuse_fast = (g_kfuse_lk % 4 == 0) && !g_force_slow;
  }
  return dispatch<ConvRun>(N, dtype, a);
}


// HBM-level outer pass (ffc_big.h).  fwd: in = long side (Bp_valid rows, Hin, Llong), out = (2*npair, Hin*N0, Mi).
int ffcsim_big_outer_r(int N0, int R, int c, int dtype, int fwd, const void* in, void* out, const void* gate, int Bp_valid, int npair,
                       int Hin, int Mi, int Llong, float scale);
int ffcsim_big_outer(int N0, int dtype, int fwd, const void* in, void* out, const void* gate, int Bp_valid, int npair,
                     int Hin, int Mi, int Llong, float scale) {
  return ffcsim_big_outer_r(N0, 1, 0, dtype, fwd, in, out, gate, Bp_valid, npair, Hin, Mi, Llong, scale);
}
// R > 1: pass c of a factor R * 32 (ffc_outer_pass_r; tables of the R-pass plan of the fused 32768 kernel)
int ffcsim_big_outer_r(int N0, int R, int c, int dtype, int fwd, const void* in, void* out, const void* gate, int Bp_valid, int npair,
                       int Hin, int Mi, int Llong, float scale) {
  HostPlan p;
  if (R > 1 && N0 != 32) return -3;
  // the library's dtype flags (ffc_k_big.hip decode_dtype): | 16 = fp32 long side, bits 8..15 = log2 of the forward prescale
  const int lf32 = (dtype & 16) ? 1 : 0;
  const int half = (dtype & 32) ? 1 : 0;      // FFC_HALF_ROWS (BigArgs::half)
  const float lpre = (float)(1u << ((dtype >> 8) & 0xff));
  dtype &= 15;
  if (!build_plan(R > 1 ? 32768 * R : (N0 == 16 ? 16384 : 32768), dtype, &p)) return -1;   // only for the N0-point operand table
  if (half && (Bp_valid != 1 || npair != 1)) return -4;
  BigArgs a{};
  a.R = R; a.c = c;
  a.lf32 = lf32; a.lpre = lpre; a.half = half;
  a.in = in; a.out = out; a.gate = gate; a.fmat = p.blob.data() + (R > 1 ? p.tabs.matk[c][fwd ? 0 : 1] : p.tabs.mat[0]);
  a.Bp_valid = Bp_valid; a.npair = npair; a.Hin = Hin; a.Mi = Mi; a.Llong = Llong; a.scale = scale;
  a.fast = (Llong % 8 == 0) && !g_force_slow;
  const int cols = N0 == 16 ? GeoBig<16>::Mi : GeoBig<32>::Mi;
  if (Mi % cols) return -2;
  const int nwg = npair * Hin * (Mi / cols);
  // the launcher's rule (ffc_k_big.hip launch_level): persistent double-buffered form for plain levels with 16-byte accesses and no
  // input gate; here 3 "workgroups" walk the blocks so that every one of them runs several iterations
  const bool pipe = g_big_pipe && R == 1 && a.fast && !a.lf32 && !a.half && !(fwd && gate);
  const int npw = pipe ? (nwg < 3 ? nwg : 3) : nwg;
  for (int wg = 0; wg < npw; wg++) {
#define FFC_BIG(NN, DD, FF) run_wg(GeoBig<NN>::WGW + (pipe ? 1 : 0), pipe ? BigBody<SimB, NN, DD>::PIPE_LDS : GeoBig<NN>::LDS_BYTES, [&]() { \
      if (pipe) BigBody<SimB, NN, DD>::template run_pipe<FF>(a, wg, npw); else BigBody<SimB, NN, DD>::template run<FF>(a, wg); })
    if (N0 == 16) { if (dtype == DT_BF16) { if (fwd) FFC_BIG(16, DT_BF16, true); else FFC_BIG(16, DT_BF16, false); }
                    else { if (fwd) FFC_BIG(16, DT_F16, true); else FFC_BIG(16, DT_F16, false); } }
    else { if (dtype == DT_BF16) { if (fwd) FFC_BIG(32, DT_BF16, true); else FFC_BIG(32, DT_BF16, false); }
           else { if (fwd) FFC_BIG(32, DT_F16, true); else FFC_BIG(32, DT_F16, false); } }
#undef FFC_BIG
  }
  return 0;
}

// all R passes in one workgroup run (ffc_outer_pass_all, BigBody::run_all)
int ffcsim_big_outer_all(int R, int dtype, int fwd, const void* in, void* out, const void* gate, int Bp_valid, int npair,
                         int Hin, int Mi, int Llong, float scale) {
  HostPlan p;
  const int lf32 = (dtype & 16) ? 1 : 0;
  const int half = (dtype & 32) ? 1 : 0;
  const float lpre = (float)(1u << ((dtype >> 8) & 0xff));
  dtype &= 15;
  if (R < 2 || R > 4 || !build_plan(32768 * R, dtype, &p)) return -1;
  if (half && (Bp_valid != 1 || npair != 1)) return -4;
  BigArgs a{};
  a.R = R; a.c = 0;
  a.lf32 = lf32; a.lpre = lpre; a.half = half;
  a.in = in; a.out = out; a.gate = gate;
  for (int c = 0; c < R; c++) a.fmats[c] = p.blob.data() + p.tabs.matk[c][fwd ? 0 : 1];
  a.fmat = a.fmats[0];
  a.Bp_valid = Bp_valid; a.npair = npair; a.Hin = Hin; a.Mi = Mi; a.Llong = Llong; a.scale = scale;
  a.fast = (Llong % 8 == 0) && !g_force_slow;
  if (Mi % GeoBig<32>::Mi) return -2;
  if (Llong > R * 32 * Mi) return -5;
  if (Llong > 32 * Mi) {      // the wide form (ffc_outer_pass_all: long side beyond the first 32 rows, BigBody::run_wide)
    if (lf32 || half || GeoBig<32>::WGW % R) return -6;
    a.wide = 1;
    const int nw = npair * Hin * (Mi / (128 * (GeoBig<32>::WGW / R)));
    for (int wg = 0; wg < nw; wg++) {
#define FFC_BIGW(DD, FF) run_wg(GeoBig<32>::WGW, GeoBig<32>::LDS_BYTES, [&]() { BigBody<SimB, 32, DD>::template run_wide<FF>(a, wg); })
      if (dtype == DT_BF16) { if (fwd) FFC_BIGW(DT_BF16, true); else FFC_BIGW(DT_BF16, false); }
      else { if (fwd) FFC_BIGW(DT_F16, true); else FFC_BIGW(DT_F16, false); }
#undef FFC_BIGW
    }
    return 0;
  }
  const int nwg = npair * Hin * (Mi / GeoBig<32>::Mi);
  for (int wg = 0; wg < nwg; wg++) {
#define FFC_BIGA(DD, FF) run_wg(GeoBig<32>::WGW, GeoBig<32>::LDS_BYTES, [&]() { BigBody<SimB, 32, DD>::template run_all<FF>(a, wg); })
    if (dtype == DT_BF16) { if (fwd) FFC_BIGA(DT_BF16, true); else FFC_BIGA(DT_BF16, false); }
    else { if (fwd) FFC_BIGA(DT_F16, true); else FFC_BIGA(DT_F16, false); }
#undef FFC_BIGA
  }
  return 0;
}

int ffcsim_upw(int N) { int u = 0; dispatch<UpwGet>(N, 0, &u); return u; }

int ffcsim_kernel_fft_c(int N, int dtype, const void* xpair, int H, void* kf, float scale) {
  HostPlan p;
  if (!build_plan(N, dtype, &p)) return -1;
  KfArgs a{};
  a.xpair = xpair; a.kf = kf; a.tab = p.blob.data(); a.t = p.tabs; a.H = H; a.Lk = N; a.scale = scale; a.prescale = 1.f; a.s_fwd = (float)p.s_fwd; a.fast = 1;
  a.R = p.R;
  return dispatch<KfRun>(N, dtype, a);
}

int ffcsim_kernel_fft(int N, int dtype, const float* k, int H, int Lk, void* kf) {
  HostPlan p;
  if (!build_plan(N, dtype, &p)) return -1;
  KfArgs a{};
  a.k = k; a.kf = kf; a.tab = p.blob.data(); a.t = p.tabs; a.H = H; a.Lk = Lk;
  a.s_fwd = (float)p.s_fwd;
  a.prescale = dtype == DT_F16 ? 256.f : 1.f;
  a.scale = (float)(p.s_k / p.s_fwd) / a.prescale; a.fast = (Lk % 4 == 0) && !g_force_slow;
  a.R = p.R;
  return dispatch<KfRun>(N, dtype, a);
}

// ws must hold nchunk*UPW*H*NT*2048 floats
int ffcsim_conv_bwd_dkf(int N, int dtype, const void* dout, const void* u, const void* pregate, const void* postgate,
                        float* ws, int B, int H, int L, int nchunk) {
  HostPlan p;
  if (!build_plan(N, dtype, &p)) return -1;
  DkfArgs d{};
  ConvArgs& a = d.c;
  a.u = u; a.pregate = pregate; a.postgate = postgate; a.tab = p.blob.data(); a.t = p.tabs;
  a.B = B; a.H = H; a.L = L; a.npair = (B + 1) / 2; a.s_fwd = (float)p.s_fwd;
  a.sbu = a.sbg = a.sbp = a.sby = (int64_t)H * L; d.sbd = d.sbdu = d.sbdpre = d.sbdpost = (int64_t)H * L;
  int upw = ffcsim_upw(N);
  int per_iter = p.N1 > 1 ? upw : upw * p.G;
  int iters_total = (a.npair + per_iter - 1) / per_iter;
  if (nchunk > iters_total) nchunk = iters_total;
  int ipc = (iters_total + nchunk - 1) / nchunk;
  a.ppc = ipc * per_iter; a.nchunk = (a.npair + a.ppc - 1) / a.ppc;
  a.fast = (L % 8 == 0) && !g_force_slow;
  a.R = p.R;
  d.dout = dout; d.ws = ws;
  std::vector<uint8_t> zs((size_t)H * a.nchunk * upw * N * 4 + 16);
  d.zscratch = zs.data();
  int rc = dispatch<DkfRun>(N, dtype, d);
  return rc < 0 ? rc : a.nchunk;   // number of slabs written (one per chunk)
}

// fused backward: du (and dpre if non-null) + dk_f slabs; returns the number of slabs
int ffcsim_conv_bwd(int N, int dtype, const void* dout, const void* u, const void* kf, const void* pregate, const void* postgate,
                    void* du, void* dpre, void* dpost, float* ws, int B, int H, int L, int nchunk) {
  HostPlan p;
  if (!build_plan(N, dtype, &p)) return -1;
  DkfArgs d{};
  ConvArgs& a = d.c;
  a.u = u; a.kf = kf; a.pregate = pregate; a.postgate = postgate; a.tab = p.blob.data(); a.t = p.tabs;
  a.B = B; a.H = H; a.L = L; a.npair = (B + 1) / 2; a.s_inv = (float)p.s_inv; a.s_fwd = (float)p.s_fwd;
  a.sbu = a.sbg = a.sbp = a.sby = (int64_t)H * L; d.sbd = d.sbdu = d.sbdpre = d.sbdpost = (int64_t)H * L;
  int upw = ffcsim_upw(N);
  int per_iter = p.N1 > 1 ? upw : upw * p.G;
  int iters_total = (a.npair + per_iter - 1) / per_iter;
  if (nchunk > iters_total) nchunk = iters_total;
  int ipc = (iters_total + nchunk - 1) / nchunk;
  a.ppc = ipc * per_iter; a.nchunk = (a.npair + a.ppc - 1) / a.ppc;
  a.fast = (L % 8 == 0) && !g_force_slow;
  a.R = p.R;
  d.dout = dout; d.ws = ws; d.du = du; d.dpre = dpre; d.dpost = (p.N1 > 1 || g_yraw) ? dpost : nullptr;
  if (p.N1 > 1 && p.R == 1 && g_z) { d.zin = g_z; d.yraw = g_yraw; a.flags = g_flags; a.stream = 1; }
  if (p.N1 <= 1) { d.zin = g_z; d.yraw = dpost ? g_yraw : nullptr; }      // (conv_bwd_impl: y_raw with or without the spectra)
  HostPlan pbf;
  if (g_dk_pair && a.nchunk == 1 && N >= 8192 && N <= 32768) {
    d.dk_pair = g_dk_pair; d.Lk = N; d.dk_scale = g_dk_pair_scale; d.dk_fast = 1;
    if (!build_plan(N, DT_BF16, &pbf)) return -1;
    d.tab_bf = pbf.blob.data(); d.t_bf = pbf.tabs;
  } else if (g_dk_out && a.nchunk == 1 && ((N >= 8192 && N <= 32768) || (p.R > 1 && p.N1 > 1 && dtype == DT_BF16))) {
    d.dk_out = g_dk_out; d.Lk = g_dk_lk; d.dk_scale = (float)(1.0 / p.s_fwd); d.dk_fast = (g_dk_lk % 4 == 0) && !g_force_slow;
    if (!build_plan(N, DT_BF16, &pbf)) return -1;
    d.tab_bf = pbf.blob.data(); d.t_bf = pbf.tabs;
  }
  std::vector<uint8_t> zs((size_t)H * a.nchunk * upw * N * 4 + 16);
  d.zscratch = zs.data();
  int rc = dispatch<BwdRun>(N, dtype, d);
  return rc < 0 ? rc : a.nchunk;
}

int ffcsim_kernel_ifft_grad_c(int N, const float* ws, int nslab, int H, void* outpair, float scale) {
  HostPlan p;
  if (!build_plan(N, DT_BF16, &p)) return -1;
  DkArgs a{};
  a.ws = ws; a.outpair = outpair; a.tab = p.blob.data(); a.t = p.tabs; a.H = H; a.Lk = N; a.nslab = nslab; a.scale = scale; a.s_inv = (float)p.s_inv; a.fast = 1;
  a.R = p.R;
  return dispatch<DkRun>(N, DT_BF16, a);
}

int ffcsim_kernel_ifft_grad(int N, int dtype, const float* ws, int nslab, int H, int Lk, float* dk) {
  HostPlan p;
  (void)dtype;                      // the dk inverse always runs in bf16 arithmetic (see ffc_k_dk.hip)
  dtype = DT_BF16;
  if (!build_plan(N, dtype, &p)) return -1;
  DkArgs a{};
  a.ws = ws; a.dk = dk; a.tab = p.blob.data(); a.t = p.tabs; a.H = H; a.Lk = Lk; a.nslab = nslab;
  a.scale = (float)(1.0 / p.s_fwd); a.s_inv = (float)p.s_inv;   // tile_inv applies s_inv = 1/(N s_fwd)
  a.fast = (Lk % 4 == 0) && !g_force_slow;
  a.R = p.R;
  return dispatch<DkRun>(N, dtype, a);
}

}  // extern "C"
