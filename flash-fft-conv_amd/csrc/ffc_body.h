// Fused Monarch FFT-convolution body, written once against a "wave backend" B.
//   * ffc_hip.hip instantiates it with the gfx950 device backend (one value per lane, MFMA /
//     LDS / global builtins).
//   * ffc_sim.cpp instantiates it with a 64-lane vector backend that executes the same code on
//     the CPU (test infrastructure: lets the index algebra be verified without a GPU).
//
// One unit = one complex sequence z = x_a + i*x_b (two batch rows that share the head h; valid
// because k is real, so Re/Im of the circular convolution are the two real outputs).
//   phase A  outer forward DFT over the strided digit n1  (global -> LDS exchange buffer E)
//   phase B  per 32x32 tile, all in registers: DFT n2, twiddle, DFT n3, (x) k_f, iDFT k3,
//            twiddle, iDFT k2, outer inverse twiddle                         (E -> E in place)
//   phase C  outer inverse DFT over k1                                       (E -> global)
// For N <= 1024 there is no outer digit: phases A/C are plain copies and a tile holds G pairs.
// Replaces the reference kernels monarch_cuda/kernels_{bf16,fp16}/monarch_cuda_*_kernel*.h
// (fwd) and, with conj_kf, the dx half of the *_bwd_kernel*.h files.
#pragma once
#include <stdint.h>

#include "ffc_layout.h"
#include "ffc_plan.h"

// activation rows go through non-temporal (streaming) global accesses: see DevB::g_r128_nt
#ifndef FFC_STREAM_ROWS
#define FFC_STREAM_ROWS 1
#endif
#ifndef FFC_FN
#define FFC_FN inline __attribute__((always_inline))
#endif
// saved spectra (ConvArgs::zsave) written / read with streaming accesses (A/B: -DFFC_Z_STREAM=false keeps them cacheable)
#ifndef FFC_Z_STREAM
#define FFC_Z_STREAM true
#endif

namespace ffc {

struct ConvArgs {
  const void* u;         // (B,H,L) dtype
  const void* pregate;   // (B,H,L) dtype or null
  const void* postgate;  // (B,H,L) dtype or null
  void* y;               // (B,H,L) dtype
  const void* kf;        // (H, NT*1024, 2) dtype, internal order, pre-scaled by s_k
  const uint8_t* tab;    // plan blob
  PlanTabs t;
  int B, H, L;
  int npair;             // ceil(B/2)
  int nchunk, ppc;       // chunks of pairs per head, pairs per chunk
  int conj_kf;           // 1: multiply by conj(k_f)  (input-gradient pass)
  int fast;              // 1: L % 8 == 0 and 16-byte aligned tensors -> 16-byte global accesses
  float s_inv;           // 1/(N*s_fwd), applied with the outer inverse twiddle (fused sizes >= 4096)
  float s_fwd;           // forward scale, applied with the outer forward twiddle (fused sizes >= 4096)
  int flags;             // reserved tuning flags
  int stream;            // 1: activation rows through non-temporal accesses (set by the host when every row is touched once)
  int persist;           // grid cap of the persistent one-wave-per-unit kernels (CU count rounded down to 8)
  int R;                 // > 1: multi-pass size, fft size = R * GEO::N (HostPlan::R); 0 / 1: single pass
  // batch strides in elements of u / pregate / postgate / y (row (b, h) starts at b * stride + h * L): H * L for contiguous
  // tensors; larger when the tensor is a channel slice of a wider (B, C, L) tensor (FlashHyenaOp: x1, x2, v are slices of the
  // short convolution's (B, 3D, L) output and are read in place)
  int64_t sbu, sbg, sbp, sby;
  // forward only, optional: the spectrum of every pair (FFT(u * pregate), dtype, internal order, N complex values per
  // (head, pair): [H][npair][N][2]) is stored here for the backward pass, which then skips its first transform
  // (ffc_conv_fwd_z / ffc_conv_bwd_z).  Fused single-pass sizes with an outer digit (fft 4096 ... 32768) only.
  void* zsave;
  // with zsave, optional: the output BEFORE the postgate multiply, contiguous (B,H,L) dtype (the gated backward's
  // dpostgate = dout * this, which then needs no inverse transform of the saved spectrum)
  void* yraw;
  // > 0: low-pass k_f whose non-zero bins all have k3 < sparse or k3 >= 32 - sparse (sparse <= 4): the kernel skips the
  // all-zero spectrum rows (ffc_conv_fwd_sparse; 32-point inner digits only)
  int sparse;
  unsigned long long* prof;  // profiling build only: per-wave phase cycle sums [wg][wave][8]
  // optional side product of the row load (rows_store / rows_in_rp): aux_out[b,h,n] = u[b,h,n] * aux_in[b,h,n], the raw row
  // before the pregate multiply.  The gated backward on saved spectra (ffc_conv_bwd_zy) reads dout as `u` here, with
  // aux_in = the forward output before the postgate multiply: aux_out = dpostgate, from the registers that hold dout anyway
  // (round 3 ran a torch elementwise kernel for it: 3 x |u| bytes of extra traffic).  Batch strides in elements.
  const void* aux_in;
  void* aux_out;
  int64_t sbai, sbao;
  // optional (round 4; ffc_conv_fwd_k, single-pass fft 32768, one chunk per head): k (H, kfuse_Lk) fp32.  The workgroup first
  // transforms ITS head's filter into `kf` (Modes::kfft_head: the k -> k_f kernel's work for one head, then phase B reads the tiles
  // the same wave stored), so the forward needs no separate k -> k_f launch.  kf stays an output: the backward pass reads it.
  const float* kfuse_k;
  const void* kfuse_x;     // instead of kfuse_k: complex input, pair-plane tensor (2, H, N) dtype (inner k_f rows of the HBM-level sizes)
  int kfuse_Lk, kfuse_fast;
  float kfuse_scale;       // s_k / s_fwd (KfArgs::scale; bf16 plans: prescale 1)
};

// One pass of a multi-pass size (fft size N = R * M, M = GEO::N = N1 * Mi; HostPlan::R).  With n = n0 M + n1 Mi + mi and
// f = k0 + R f', pass k0 computes the M-point circular convolution of
//   x_k0[n1 Mi + mi] = W_N^{(n1 Mi + mi) k0} * sum_n0 W_R^{n0 k0} z[n0 M + n1 Mi + mi]
// with the kernel samples K[k0 + R f'], and the result is  y[n0 M + m] = (1/R) sum_k0 W_R^{-n0 k0} W_N^{-m k0} y_k0[m].
// Inside the fused kernel this costs almost nothing: the factor W_{R N1}^{n1 k0} is folded into the outer digit's DFT
// matrix (plan tables matk[k0]), the factor W_N^{mi k0} into the phase of the outer twiddle chain (k1 -> k0 + R k1 on an
// N-point circle), the sum over n0 is a +-1 / +-i combination of input rows (absent when L <= M), and the sum over k0 is a
// read-modify-write of the output rows by the same wave (absent for k0 = 0).  The reference reaches these sizes through
// an extra HBM round trip (butterfly kernels, csrc/flashfftconv/butterfly/*).
struct Pass {
  int k0 = 0, R = 1;
  const uint8_t* mat_fwd = nullptr;   // outer-digit operand tables of this pass (global memory)
  const uint8_t* mat_inv = nullptr;
};

template <class B, class GEO, int DT>
struct Body {
  using f32 = typename B::f32;
  using i32 = typename B::i32;
  using u32 = typename B::u32;
  using pred = typename B::pred;
  using U2 = typename B::U2;
  using U4 = typename B::U4;
  using A16 = typename B::A16;   // MFMA accumulator: 16 fp32 per lane (a native 16-register vector on the device)
  using W4 = typename B::W4;     // MFMA operand: 8 x 16-bit per lane (4 dwords)

  struct Mat { W4 w[2][3]; };       // [K-step][Fr,Fi,-Fi]
  struct CT16 { A16 re, im; };          // 16 complex fp32 per lane, accumulator-shaped (packed-math friendly)
  struct Op { W4 r[2], i[2]; };     // complex MFMA data operand, 2 K-steps

  static FFC_FN void load_mat(Mat& m, const uint8_t* p, i32 lane) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
      U4 v = B::g_r128(p, lane + q * 64);
      m.w[q / 3][q % 3] = B::w4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int q = 0; q < 6; q++) B::pin(m.w[q / 3][q % 3]);
  }
  // the same loads without the pins: issued early (ffc_big.h run<>), consumed -- and waited for -- where the matrix is first used
  static FFC_FN void load_mat_issue(Mat& m, const uint8_t* p, i32 lane) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
      U4 v = B::g_r128(p, lane + q * 64);
      m.w[q / 3][q % 3] = B::w4(v.x, v.y, v.z, v.w);
    }
  }
  static FFC_FN void load_ct16(CT16& c, const uint8_t* p, i32 lane) {
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      U4 v = B::g_r128(p, lane + rr * 64);
      c.re[2 * rr] = B::as_f32(v.x); c.re[2 * rr + 1] = B::as_f32(v.y);
      c.im[2 * rr] = B::as_f32(v.z); c.im[2 * rr + 1] = B::as_f32(v.w);
    }
  }
  // acc += M (x) data.  AFORM: data is the MFMA A operand (its lane index moves to registers,
  // the transformed index lands on lanes).  !AFORM: matrix is A (transformed index in registers,
  // data lane index stays on lanes).  CONJ selects the inverse DFT.  ms_lim: K-steps to run.
  template <bool CONJ, bool AFORM>
  static FFC_FN void cmm(A16& ore, A16& oim, const Op& d, const Mat& F, int ms_lim = 2) {
#pragma unroll
    for (int ms = 0; ms < 2; ms++) {
      if (ms >= ms_lim) continue;
      const W4& fr = F.w[ms][0];
      const W4& fi_re = F.w[ms][CONJ ? 1 : 2];  // multiplies data.im into re
      const W4& fi_im = F.w[ms][CONJ ? 2 : 1];  // multiplies data.re into im
      if (AFORM) {
        B::template mfma<DT>(ore, d.r[ms], fr);
        B::template mfma<DT>(oim, d.r[ms], fi_im);
        B::template mfma<DT>(ore, d.i[ms], fi_re);
        B::template mfma<DT>(oim, d.i[ms], fr);
      } else {
        B::template mfma<DT>(ore, fr, d.r[ms]);
        B::template mfma<DT>(oim, fi_im, d.r[ms]);
        B::template mfma<DT>(ore, fi_re, d.i[ms]);
        B::template mfma<DT>(oim, fr, d.i[ms]);
      }
    }
  }
  static FFC_FN void to_op(const A16& re, const A16& im, Op& o) {
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        o.r[ms][d] = B::template pack<DT>(re[8 * ms + 2 * d], re[8 * ms + 2 * d + 1]);
        o.i[ms][d] = B::template pack<DT>(im[8 * ms + 2 * d], im[8 * ms + 2 * d + 1]);
      }
  }
  // Folded-twiddle matrices (FFC_FOLD_TW): only the (Fr, Fi) forms are loaded (16 registers, 4 KB of L2 traffic instead of 24 / 6 KB);
  // the -Fi product is Fi times the negated imaginary operand (sign bits flipped: 4 v_xor per K-step)
  struct Mat2 { W4 w[2][2]; };
  static FFC_FN void load_mat2_issue(Mat2& m, const uint8_t* p, i32 lane) {
#if defined(FFC_KO) && (FFC_KO & 2048)
    // knock-out timing experiment (results wrong): no L2 traffic for the folded matrices -- every "load" is the lane id, so that the
    // folded kernels' instruction stream can be timed without the cost of fetching the tables
    { const u32 z = B::as_u32(B::i2f(lane));
      for (int ms = 0; ms < 2; ms++) for (int f = 0; f < 2; f++) m.w[ms][f] = B::w4(z, z, z, z);
      (void)p; return; }
#endif
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
      for (int f = 0; f < 2; f++) {
        U4 v = B::g_r128(p, lane + (ms * 3 + f) * 64);
        m.w[ms][f] = B::w4(v.x, v.y, v.z, v.w);
      }
  }
  static FFC_FN W4 neg4(const W4& x) {
    return B::w4(x[0] ^ B::uconst(0x80008000u), x[1] ^ B::uconst(0x80008000u), x[2] ^ B::uconst(0x80008000u), x[3] ^ B::uconst(0x80008000u));
  }
  // acc += F (x) data for a full complex matrix F = Fr + i Fi (conjugation, if any, is in the table)
  template <bool AFORM>
  static FFC_FN void cmm2(A16& ore, A16& oim, const Op& d, const Mat2& F) {
#pragma unroll
    for (int ms = 0; ms < 2; ms++) {
      const W4 ni = neg4(d.i[ms]);
      if (AFORM) {
        B::template mfma<DT>(ore, d.r[ms], F.w[ms][0]);
        B::template mfma<DT>(oim, d.r[ms], F.w[ms][1]);
        B::template mfma<DT>(ore, ni, F.w[ms][1]);
        B::template mfma<DT>(oim, d.i[ms], F.w[ms][0]);
      } else {
        B::template mfma<DT>(ore, F.w[ms][0], d.r[ms]);
        B::template mfma<DT>(oim, F.w[ms][1], d.r[ms]);
        B::template mfma<DT>(ore, F.w[ms][1], ni);
        B::template mfma<DT>(oim, F.w[ms][0], d.i[ms]);
      }
    }
  }
  // element-wise complex multiplies on whole accumulator tuples: the device backend issues them as packed
  // fp32 math (v_pk_mul_f32 / v_pk_fma_f32 on register pairs = half the VALU issue slots).
  static FFC_FN void cmul(A16& re, A16& im, const CT16& t) { B::template cmul16<false>(re, im, t.re, t.im); }
  static FFC_FN void cmul_conj(A16& re, A16& im, const CT16& t) { B::template cmul16<true>(re, im, t.re, t.im); }
  // Twiddle chain: t[i] = scale * cis(sign * 2*pi * ((base + off_i*step) mod N) / N) for the 8 row offsets
  // off = {0,1,2,3,8,9,10,11} an accumulator half holds: three v_sin/v_cos pairs (exact integer phases,
  // argument in revolutions) and seven complex multiplies instead of 8 sincos + per-element phase math.
  // nmask / inv_n: the circle the phase lives on (N - 1, 1 / N); compile-time constants for the single-pass sizes
  static FFC_FN void cis_rev(i32 phase, float sign, f32* c, f32* s, int nmask = GEO::N - 1, float inv_n = 1.0f / (float)GEO::N) {
    f32 x = B::i2f(phase & nmask) * inv_n;
    f32 cc = B::cos_rev(x), ss = B::sin_rev(x);
    B::settle(cc, ss);     // see DevB::settle: transcendental results are fenced before packed-math consumers
    *c = cc;
    *s = ss * sign;
  }
  static FFC_FN void chain8(i32 base, i32 step, float sign, float scale, f32 (&tr)[8], f32 (&ti)[8]) {
    f32 c0, s0, c1, s1, c8, s8;
    cis_rev(base, sign, &c0, &s0);
    cis_rev(step, sign, &c1, &s1);
    cis_rev(step * 8, sign, &c8, &s8);
    tr[0] = c0 * scale; ti[0] = s0 * scale;
    tr[4] = tr[0] * c8 - ti[0] * s8; ti[4] = tr[0] * s8 + ti[0] * c8;
#pragma unroll
    for (int i = 1; i < 4; i++) {
      tr[i] = tr[i - 1] * c1 - ti[i - 1] * s1; ti[i] = tr[i - 1] * s1 + ti[i - 1] * c1;
      tr[4 + i] = tr[3 + i] * c1 - ti[3 + i] * s1; ti[4 + i] = tr[3 + i] * s1 + ti[3 + i] * c1;
    }
  }
  // Same chain with packed math: pairs of consecutive rows {0,1},{2,3},{8,9},{10,11}; t1 = t0 w, then the pairs
  // are stepped by w^2 and w^8 (two elements per packed instruction).
  using F2 = typename B::F2;
  static FFC_FN void chain8p(i32 base, i32 step, float sign, float scale, F2 (&tr)[4], F2 (&ti)[4],
                             int nmask = GEO::N - 1, float inv_n = 1.0f / (float)GEO::N, f32* w8c = nullptr, f32* w8s = nullptr) {
    f32 c0, s0, c1, s1, c8, s8;
    cis_rev(base, sign, &c0, &s0, nmask, inv_n);
    cis_rev(step, sign, &c1, &s1, nmask, inv_n);
    cis_rev(step * 8, sign, &c8, &s8, nmask, inv_n);
    c0 = c0 * scale; s0 = s0 * scale;
    f32 c2 = c1 * c1 - s1 * s1, s2 = (c1 + c1) * s1;
    tr[0] = B::f2(c0, c0 * c1 - s0 * s1);
    ti[0] = B::f2(s0, c0 * s1 + s0 * c1);
    B::cmulp(tr[0], ti[0], c2, s2, tr[1], ti[1]);
    B::cmulp(tr[0], ti[0], c8, s8, tr[2], ti[2]);
    B::cmulp(tr[1], ti[1], c8, s8, tr[3], ti[3]);
    if (w8c) { *w8c = c8; *w8s = s8; }
  }
  // apply a chain8p result to accumulator rows 8*half + {0..7}
  static FFC_FN void apply8(A16& re, A16& im, int half, const F2 (&tr)[4], const F2 (&ti)[4]) {
#if defined(FFC_KO) && (FFC_KO & 1)
    return;      // timing experiment only: results are wrong
#endif
#pragma unroll
    for (int i = 0; i < 4; i++) B::template cmul2v<false>(re, im, 8 * half + 2 * i, tr[i], ti[i]);
  }
  // Round 5: both accumulator halves of a tile from ONE chain when they share the step (rows off and off + 16 of the same
  // twiddle column: 32-point outer digit in phases A, 32-point last inner digit in the inverse twiddle).  The second half is the
  // first one times w^16 = (w^8)^2 -- four packed complex multiplies instead of a second chain with its three v_sin / v_cos pairs,
  // their argument preparation and the wait states behind them (6 -> 3 transcendental pairs per tile).
  // FFC_CHAIN16=0: two chains, the round-4 form (A/B builds).
#ifndef FFC_CHAIN16
#define FFC_CHAIN16 1
#endif
  // phases A: 32-point outer digit; not in the 128-VGPR kernels of fft 8192 (Geo<32,16,16> on a LEAN_OUTER backend: with the chain's
  // w^8 kept for the second half the allocator parked values in a0..a3 -- build.py check_agpr)
  static constexpr bool CHAIN16_A = GEO::N1 == 32 && (!B::LEAN_OUTER || GEO::N2 == 32);
  // the per-tile outer stage (128-VGPR kernels, full-length rows) keeps two chains: with one, the recomputing full-length backward of
  // fft 32768 spilled one value into a0 (build.py check_agpr)
#ifndef FFC_CHAIN16_TILE
#define FFC_CHAIN16_TILE 0
#endif
  static FFC_FN void twiddle16(A16& re, A16& im, i32 base, i32 step, float sign, float scale,
                               int nmask = GEO::N - 1, float inv_n = 1.0f / (float)GEO::N) {
    F2 tr[4], ti[4];
    if constexpr (FFC_CHAIN16 != 0) {
      f32 c8, s8;
      chain8p(base, step, sign, scale, tr, ti, nmask, inv_n, &c8, &s8);
      apply8(re, im, 0, tr, ti);
      const f32 c16 = c8 * c8 - s8 * s8, s16 = (c8 + c8) * s8;
#pragma unroll
      for (int i = 0; i < 4; i++) B::cmulp(tr[i], ti[i], c16, s16, tr[i], ti[i]);
      apply8(re, im, 1, tr, ti);
    } else {
      chain8p(base, step, sign, scale, tr, ti, nmask, inv_n);
      apply8(re, im, 0, tr, ti);
      chain8p(base + 16 * step, step, sign, scale, tr, ti, nmask, inv_n);
      apply8(re, im, 1, tr, ti);
    }
  }
#ifndef FFC_PK_GATE
#define FFC_PK_GATE 1
#endif
  // dtype pair (x) dtype pair, rounded back to dtype (the reference multiplies gates in the
  // activation dtype: kernels_bf16/monarch_cuda_32_32_32_kernel_bf16.h:409-429, 613-634).
  static FFC_FN u32 mul2(u32 a, u32 g) {
    // fp16: one v_pk_mul_f16 (round 5).  The product of two fp16 values is exact in fp32, so "fp32 product rounded once" IS the native
    // fp16 multiply -- bit for bit what the widening path below computes (2 x 2 conversions, 2 multiplies, 1 pack), and what the
    // reference's __hmul2 does.  bf16 has no packed multiply on gfx950.  FFC_PK_GATE=0: the widening path for both dtypes (A/B builds)
    if constexpr (DT == DT_F16 && FFC_PK_GATE != 0) return B::pk_mul_f16(a, g);
    f32 lo = B::template unpack_lo<DT>(a) * B::template unpack_lo<DT>(g);
    f32 hi = B::template unpack_hi<DT>(a) * B::template unpack_hi<DT>(g);
    return B::template pack<DT>(lo, hi);
  }
  // raw[e] holds 4 consecutive columns (tiles t=0..3) of k-slot e; build tile t's operand dwords.
  static FFC_FN void xpose(const U2 (&raw)[8], int t, W4& op) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
      u32 a = (t < 2) ? raw[2 * d].x : raw[2 * d].y;
      u32 b = (t < 2) ? raw[2 * d + 1].x : raw[2 * d + 1].y;
      op[d] = (t & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
  }

  // same with a runtime tile pair tp (x or y dword) and compile-time half th
  static FFC_FN void xpose2(const U2 (&raw)[8], int tp, int th, W4& op) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
      u32 a = tp ? raw[2 * d].y : raw[2 * d].x;
      u32 b = tp ? raw[2 * d + 1].y : raw[2 * d + 1].x;
      op[d] = th ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
  }

  // ------------------------------------------------------------------ LDS-resident tables
  static FFC_FN void copy_tab(const uint8_t* src, int lds_off, int bytes) {
    const i32 tid = B::lane() + B::wave() * 64;
    for (int i = 0; i < bytes / 16; i += GEO::WGW * 64) {
      i32 idx = tid + i;
      pred ok = idx < bytes / 16;
      U4 v = B::g_r128p(src, idx, ok);
      B::lds_w128(idx * 16 + lds_off, v, ok);
    }
  }
  // table pieces of this thread: loads from a clamped index (always issued, no exec-masked branch), LDS writes under the predicate
  static constexpr int TAB_T = GEO::WGW * 64;
  template <int BYTES>
  static FFC_FN void tab_issue(const uint8_t* src, U4 (&v)[(BYTES / 16 + TAB_T - 1) / TAB_T]) {
    const i32 tid = B::lane() + B::wave() * 64;
#pragma unroll
    for (int it = 0; it < (BYTES / 16 + TAB_T - 1) / TAB_T; it++) v[it] = B::g_r128(src, B::imin(tid + it * TAB_T, BYTES / 16 - 1));
  }
  template <int BYTES>
  static FFC_FN void tab_commit(int lds_off, const U4 (&v)[(BYTES / 16 + TAB_T - 1) / TAB_T]) {
    const i32 tid = B::lane() + B::wave() * 64;
#pragma unroll
    for (int it = 0; it < (BYTES / 16 + TAB_T - 1) / TAB_T; it++) {
      i32 idx = tid + it * TAB_T;
      B::lds_w128(idx * 16 + lds_off, v[it], idx < BYTES / 16);
    }
  }
  // Round 4: every table's loads are requested first, then written to LDS (copy_tab one table after the other put each load into
  // its own branch followed by s_waitcnt vmcnt(0): 3 .. 6 L2 round trips in a row at the start of every workgroup -- a tenth of the
  // run time of the short-sequence launches).
  static FFC_FN void setup_tables(const uint8_t* tab, const PlanTabs& t) {
    U4 f1[(6144 / 16 + TAB_T - 1) / TAB_T], f2[(6144 / 16 + TAB_T - 1) / TAB_T], f3[(6144 / 16 + TAB_T - 1) / TAB_T],
       tw[(8192 / 16 + TAB_T - 1) / TAB_T], tw2[(8192 / 16 + TAB_T - 1) / TAB_T], fs[(3072 / 16 + TAB_T - 1) / TAB_T];
    if constexpr (GEO::OUTER) tab_issue<6144>(tab + t.mat[0], f1);
    tab_issue<6144>(tab + t.mat[1], f2);
    tab_issue<8192>(tab + t.twin, tw);
    if constexpr (GEO::N3 != GEO::N2) tab_issue<6144>(tab + t.mat[2], f3);
    if constexpr (GEO::TW2_SEP) tab_issue<8192>(tab + t.twin2, tw2);
    if constexpr (GEO::HAS_SP) tab_issue<3072>(tab + t.mat_sp, fs);
    B::sched_fence();
    if constexpr (GEO::OUTER) tab_commit<6144>(GEO::L_F1, f1);
    tab_commit<6144>(GEO::L_F2, f2);
    tab_commit<8192>(GEO::L_TW, tw);
    if constexpr (GEO::N3 != GEO::N2) tab_commit<6144>(GEO::L_F3, f3);
    if constexpr (GEO::TW2_SEP) tab_commit<8192>(GEO::L_TW2, tw2);
    if constexpr (GEO::HAS_SP) tab_commit<3072>(GEO::L_FS, fs);
    B::barrier();
  }
  static FFC_FN void lds_mat(Mat& m, int off) {
    const i32 lane = B::lane();
#pragma unroll
    for (int q = 0; q < 6; q++) {
      U4 v = B::lds_r128(lane * 16 + (off + q * 1024));
      m.w[q / 3][q % 3] = B::w4(v.x, v.y, v.z, v.w);
    }
    // after all six loads are in flight: the pin consumes its operand, i.e. waits for it
#pragma unroll
    for (int q = 0; q < 6; q++) B::pin(m.w[q / 3][q % 3]);
  }
  static FFC_FN void lds_ct16(CT16& c, int off) {
    const i32 lane = B::lane();
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      U4 v = B::lds_r128(lane * 16 + (off + rr * 1024));
      c.re[2 * rr] = B::as_f32(v.x); c.re[2 * rr + 1] = B::as_f32(v.y);
      c.im[2 * rr] = B::as_f32(v.z); c.im[2 * rr + 1] = B::as_f32(v.w);
    }
  }
  // x (x) tab (or conj tab) with the table streamed from LDS two rows at a time (4 transient registers
  // instead of a 32-register CT16)
  template <bool CONJ>
  static FFC_FN void cmul_lds(A16& re, A16& im, int off) {
    const i32 lane = B::lane();
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      U4 v = B::lds_r128(lane * 16 + (off + rr * 1024));
      B::template cmul2<CONJ>(re, im, 2 * rr, B::as_f32(v.x), B::as_f32(v.y), B::as_f32(v.z), B::as_f32(v.w));
    }
  }
  // t[r] *= tab[hi][r] where tab is a [2][16] complex f32 LDS table (lane-uniform per half-wave)
  static FFC_FN void cmul_small(CT16& t, int off) {
    const i32 hi = B::lane() >> 5;
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      U4 v = B::lds_r128(hi * 128 + (off + rr * 16));
      f32 br[2] = {B::as_f32(v.x), B::as_f32(v.z)}, bi[2] = {B::as_f32(v.y), B::as_f32(v.w)};
#pragma unroll
      for (int q = 0; q < 2; q++) {
        int r = 2 * rr + q;
        f32 x = t.re[r], y = t.im[r];
        t.re[r] = x * br[q] - y * bi[q];
        t.im[r] = x * bi[q] + y * br[q];
      }
    }
  }

  struct Unit { int eb; int wq; };   // E base (bytes) of this wave's unit, wave index inside the unit
  static FFC_FN Pass make_pass(const ConvArgs& a, int k0) {
    Pass ps;
    ps.k0 = k0; ps.R = a.R;
    ps.mat_fwd = a.tab + a.t.matk[k0][0]; ps.mat_inv = a.tab + a.t.matk[k0][1];
    return ps;
  }

  // ------------------------------------------------------------------ global <-> E row copies
  // A unit's E holds ROWS rows of Mi points per plane.  OUTER: the rows are the n1 slices of one
  // pair (plane 0 = batch row 2p, plane 1 = row 2p+1) and a wave moves only its own 128*S1-column
  // slice (the columns it transforms in phases A/C, so no barrier is needed around the copies).
  // Inner-only sizes: row g = pair q*G+g, the single wave of the unit moves the whole tile.
  // 16-byte global accesses, 1 KiB contiguous per wave instruction when L % 8 == 0.
  // Streaming rows (ConvArgs::stream, chosen by the host; FFC_STREAM=0/1 overrides it for A/B runs, benchmarks/prof_stream.py).
  // Same process, same box, B16 H768: forward 16K 0.323 -> 0.313 ms, gated forward 32K 0.442 -> 0.410; ungated backward
  // 16K 0.667 -> 0.638, 32K 1.043 -> 1.025.  Not used in the gated backward, where the same workgroup reads u, dout and
  // the gates a second time as output gates (cfg3: 0.839 -> 0.865 with streaming).
  static constexpr bool STREAM_ROWS = FFC_STREAM_ROWS != 0;
  static FFC_FN int64_t row_off(int b, bool ok, int64_t sb, int h, int L) { return (int64_t)(ok ? b : 0) * sb + (int64_t)h * L; }
  struct RowIO {
    const uint16_t* src[2]; const uint16_t* gate[2]; uint16_t* dst[2]; bool valid[2];
  };
  // fast path: an unconditional 16-byte load from a clamped (always valid) position; positions beyond L and
  // missing batch rows are zeroed when the registers are written to E (rows_store).  No branch, no
  // zero-initialised destination: consecutive loads never wait for each other, and the loads of the next pair
  // can stay in flight across phase C.
  static FFC_FN U4 gload8(const uint16_t* base, i32 n, int L, int fast, bool rowok) {
#if defined(FFC_KO) && (FFC_KO & 2)
    { U4 z; z.x = B::as_u32(B::i2f(n)); z.y = z.x; z.z = z.x; z.w = z.x; return z; }     // knock-out experiment: no row loads
#endif
    // B::FAST_ONLY: kernel instantiations that are only launched on 16-byte-aligned tensors with L % 8 == 0 (the element-wise arm is
    // compiled out: with both arms in one kernel every load sits in its own flow block with a `s_waitcnt vmcnt(0)` at the merge)
    if (B::FAST_ONLY || fast) return (STREAM_ROWS && fast == 2) ? B::g_r128_nt(base, B::imin(n, L - 8) >> 3) : B::g_r128(base, B::imin(n, L - 8) >> 3);
    u32 w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      u32 lo = B::g_r16(base, n + 2 * q, ((n + 2 * q) < L) && rowok);
      u32 hi = B::g_r16(base, n + (2 * q + 1), ((n + (2 * q + 1)) < L) && rowok);
      w[q] = lo | (hi << 16);
    }
    U4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
  }
  static FFC_FN void gstore8(uint16_t* base, i32 n, int L, int fast, bool rowok, U4 v) {
#if defined(FFC_KO) && (FFC_KO & 2)
    if (B::as_f32(v.x) != B::as_f32(v.x) + 1.0f) return;      // knock-out experiment: (practically) never stores
#endif
    if (B::FAST_ONLY || fast) {
      if (STREAM_ROWS && fast == 2) B::g_w128_nt(base, n >> 3, v, (n < L) && rowok);
      else B::g_w128(base, n >> 3, v, (n < L) && rowok);
      return;
    }
    u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      B::g_w16(base, n + 2 * q, w[q] & 0xffffu, ((n + 2 * q) < L) && rowok);
      B::g_w16(base, n + (2 * q + 1), w[q] >> 16, ((n + (2 * q + 1)) < L) && rowok);
    }
  }
  // LDS address of the 16-byte pair of chunks holding columns m..m+7 (m % 8 == 0) of `row`, and
  // whether the two 8-byte halves are swapped by the bank swizzle.
  static FFC_FN i32 pair_off(i32 row, i32 m, pred* swapped) {
    i32 n2 = m / GEO::N3, n3 = m % GEO::N3;
    i32 sig = (n2 / GEO::PER) % GEO::CR;
    *swapped = (sig & 1) >= 1;
    return row * (GEO::Mi * 2) + (n2 * GEO::CR + (((n3 >> 3) ^ (sig >> 1)) << 1)) * 8;
  }
  static constexpr int CPR = GEO::OUTER ? 16 * GEO::S1 : GEO::Mi / 8;   // 16-B chunks per row per wave
  static constexpr int NCH = GEO::OUTER ? 8 : 2;                        // chunks per lane per plane

  // NC = chunks per lane per plane actually moved: NCH, or NCH/2 when only E rows < 16 carry data
  template <int NC> struct RowRegsT { U4 v[NC][2]; };
  using RowRegs = RowRegsT<NCH>;
  // issue the global loads of pair/tile pq (no LDS access): can be overlapped with compute
  // chunks I0 .. I0+NC-1 of the wave's slice (I0 > 0: the second half of a split prefetch)
  template <int NC, int I0 = 0>
  static FFC_FN void rows_load(const ConvArgs& a, int h, int pq, Unit un, RowRegsT<NC>& X) {
    const i32 lane = B::opaque(B::lane());
    const int fast = a.fast ? (a.stream ? 2 : 1) : 0;     // 2: streaming (non-temporal) fast path
    if constexpr (!GEO::OUTER) {
      // single-tile sizes: the 16-byte / element-wise decision once for the whole tile (round 4, late).  Inside gload8_rows it put each
      // of the tile's four loads into a flow block whose merge with the element-wise arm carries a `s_waitcnt vmcnt(0)`: `L W0` four
      // times, one request per memory round trip and wave -- 16 KB in flight per CU where the memory system wants ~48 (the short
      // sequences sat at 2.6 - 3.4 TB/s).  Same loads, same predicates; only the branch moved.
      if (fast) {
#pragma unroll
        for (int ii = 0; ii < NC; ii++) {
          i32 idx = lane + (ii + I0) * 64;
          i32 row = idx / CPR, m = (idx % CPR) * 8;
#pragma unroll
          for (int pl = 0; pl < 2; pl++) {
            i32 b = (row + pq * GEO::G) * 2 + pl;
            X.v[ii][pl] = gload8_rows((const uint16_t*)a.u, b, h, a, a.sbu, m, 1, b < a.B);
          }
        }
        return;
      }
    }
#pragma unroll
    for (int ii = 0; ii < NC; ii++) {
      const int i = ii + I0;
      i32 idx = lane + i * 64;
      i32 row = idx / CPR, m = (idx % CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        if constexpr (GEO::OUTER) {
          // every lane of this chunk is beyond L (wave-uniform: the chunk's first row is a compile-time constant):
          // nothing to fetch, rows_store writes zeros (L <= N/2 on the 16-point-digit sizes skips half the loads)
          if (fast && ((i * 64) / CPR) * GEO::Mi >= a.L) { X.v[ii][pl] = U4{B::uconst(0), B::uconst(0), B::uconst(0), B::uconst(0)}; continue; }
          const int b = 2 * pq + pl;
          const bool ok = b < a.B;
          const int64_t ro = row_off(b, ok, a.sbu, h, a.L);
          X.v[ii][pl] = gload8((const uint16_t*)a.u + ro, row * GEO::Mi + m, a.L, fast, ok);
        } else {
          i32 b = (row + pq * GEO::G) * 2 + pl;
          X.v[ii][pl] = gload8_rows((const uint16_t*)a.u, b, h, a, a.sbu, m, fast, b < a.B);
        }
      }
    }
  }
  // Gated rows on the 16-byte path (round 4).  The gate (and side-product) rows used to be loaded where they are multiplied in, one
  // chunk at a time: load, s_waitcnt vmcnt(0), multiply, next chunk -- 16 memory round trips in a row per pair and wave, each
  // also waiting for every other load in flight (ISA of the gated kernels; the ungated path had its 16 row loads batched since round 1).
  // Now the loads of GATE_BATCH chunks (both planes) are issued together ahead of the work on them.  The batch is a per-
  // translation-unit constant: the forward kernels take all chunks of the wave's slice, the backward kernels (128-VGPR budget) two.
#ifndef FFC_GATE_BATCH
#define FFC_GATE_BATCH 4
#endif
  template <int NC, int I0>
  static FFC_FN void rows_store_g(const ConvArgs& a, int h, int pq, Unit un, const RowRegsT<NC>& X) {
    constexpr int GB = FFC_GATE_BATCH <= 0 ? 1 : (NC < FFC_GATE_BATCH ? NC : FFC_GATE_BATCH);
    static_assert(NC % GB == 0, "gate batch");
    const i32 lane = B::opaque(B::lane());
    const int fast = a.stream ? 2 : 1;
    const bool hasg = a.pregate != nullptr, hasa = a.aux_in != nullptr;
#pragma unroll
    for (int ib = 0; ib < NC; ib += GB) {
      U4 G[GB][2], A[GB][2];
#pragma unroll
      for (int jj = 0; jj < GB; jj++) {
        const int i = ib + jj + I0;
        i32 idx = lane + i * 64;
        i32 row = idx / CPR, m = (idx % CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
        if (GEO::OUTER && ((i * 64) / CPR) * GEO::Mi >= a.L) continue;      // chunk beyond L: zeros, nothing to gate
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
          if constexpr (GEO::OUTER) {
            const int b = 2 * pq + pl;
            const bool ok = b < a.B;
            const i32 n = row * GEO::Mi + m;
            if (hasg) G[jj][pl] = gload8((const uint16_t*)a.pregate + row_off(b, ok, a.sbg, h, a.L), n, a.L, fast, ok);
            if (hasa) A[jj][pl] = gload8((const uint16_t*)a.aux_in + row_off(b, ok, a.sbai, h, a.L), n, a.L, fast, ok);
          } else {
            i32 b = (row + pq * GEO::G) * 2 + pl;
            if (hasg) G[jj][pl] = gload8_rows((const uint16_t*)a.pregate, b, h, a, a.sbg, m, fast, b < a.B);
            if (hasa) A[jj][pl] = gload8_rows((const uint16_t*)a.aux_in, b, h, a, a.sbai, m, fast, b < a.B);
          }
        }
      }
      B::sched_fence();
#pragma unroll
      for (int jj = 0; jj < GB; jj++) {
        const int ii = ib + jj, i = ii + I0;
        i32 idx = lane + i * 64;
        i32 row = idx / CPR, m = (idx % CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
        pred sw;
        i32 off = pair_off(row, m, &sw) + un.eb;
        const bool beyond = GEO::OUTER && ((i * 64) / CPR) * GEO::Mi >= a.L;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
          U4 v = X.v[ii][pl];
          if (GEO::OUTER) {        // masks of the unconditional fast-path loads
            pred ok = ((row * GEO::Mi + m) < a.L) && ((2 * pq + pl) < a.B);
            v.x = B::sel(ok, v.x, B::uconst(0)); v.y = B::sel(ok, v.y, B::uconst(0));
            v.z = B::sel(ok, v.z, B::uconst(0)); v.w = B::sel(ok, v.w, B::uconst(0));
          }
          if (hasa && !beyond) {      // side product aux_out = row * aux_in
            if constexpr (GEO::OUTER) {
              const int b = 2 * pq + pl;
              const bool ok = b < a.B;
              gstore8((uint16_t*)a.aux_out + row_off(b, ok, a.sbao, h, a.L), row * GEO::Mi + m, a.L, fast, ok, mul4(v, A[jj][pl]));
            } else {
              i32 b = (row + pq * GEO::G) * 2 + pl;
              gstore8_rows((uint16_t*)a.aux_out, b, h, a, a.sbao, m, fast, b < a.B, mul4(v, A[jj][pl]));
            }
          }
          if (hasg && !beyond) v = mul4(v, G[jj][pl]);
          U4 o;
          o.x = B::sel(sw, v.z, v.x); o.y = B::sel(sw, v.w, v.y);
          o.z = B::sel(sw, v.x, v.z); o.w = B::sel(sw, v.y, v.w);
          B::lds_w128(off + pl * GEO::PLANE, o, B::ptrue());
        }
      }
    }
  }
  // (x) pregate, swizzle, write to E
  template <int NC, int I0 = 0>
  static FFC_FN void rows_store(const ConvArgs& a, int h, int pq, Unit un, const RowRegsT<NC>& X) {
    if constexpr (FFC_GATE_BATCH > 0) {      // (0: A/B builds of the one-load-at-a-time form below)
      if (a.fast && (a.pregate || a.aux_in)) { rows_store_g<NC, I0>(a, h, pq, un, X); return; }
    }
    const i32 lane = B::opaque(B::lane());
    const int fast = a.fast ? (a.stream ? 2 : 1) : 0;     // 2: streaming (non-temporal) fast path
#pragma unroll
    for (int ii = 0; ii < NC; ii++) {
      const int i = ii + I0;
      i32 idx = lane + i * 64;
      i32 row = idx / CPR, m = (idx % CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
      pred sw;
      i32 off = pair_off(row, m, &sw) + un.eb;
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        U4 v = X.v[ii][pl];
        if (fast && GEO::OUTER) {        // masks of the unconditional fast-path loads
          pred ok = ((row * GEO::Mi + m) < a.L) && ((2 * pq + pl) < a.B);
          v.x = B::sel(ok, v.x, B::uconst(0)); v.y = B::sel(ok, v.y, B::uconst(0));
          v.z = B::sel(ok, v.z, B::uconst(0)); v.w = B::sel(ok, v.w, B::uconst(0));
        }
        if (a.aux_in && !(GEO::OUTER && fast && ((i * 64) / CPR) * GEO::Mi >= a.L)) {      // side product aux_out = row * aux_in
          if constexpr (GEO::OUTER) {
            const int b = 2 * pq + pl;
            const bool ok = b < a.B;
            const i32 n = row * GEO::Mi + m;
            U4 r = gload8((const uint16_t*)a.aux_in + row_off(b, ok, a.sbai, h, a.L), n, a.L, fast, ok);
            gstore8((uint16_t*)a.aux_out + row_off(b, ok, a.sbao, h, a.L), n, a.L, fast, ok, mul4(v, r));
          } else {
            i32 b = (row + pq * GEO::G) * 2 + pl;
            U4 r = gload8_rows((const uint16_t*)a.aux_in, b, h, a, a.sbai, m, fast, b < a.B);
            gstore8_rows((uint16_t*)a.aux_out, b, h, a, a.sbao, m, fast, b < a.B, mul4(v, r));
          }
        }
        if (a.pregate && !(GEO::OUTER && fast && ((i * 64) / CPR) * GEO::Mi >= a.L)) {
          U4 g;
          if constexpr (GEO::OUTER) {
            const int b = 2 * pq + pl;
            const bool ok = b < a.B;
            const int64_t ro = row_off(b, ok, a.sbg, h, a.L);
            g = gload8((const uint16_t*)a.pregate + ro, row * GEO::Mi + m, a.L, fast, ok);
          } else {
            i32 b = (row + pq * GEO::G) * 2 + pl;
            g = gload8_rows((const uint16_t*)a.pregate, b, h, a, a.sbg, m, fast, b < a.B);
          }
          v.x = mul2(v.x, g.x); v.y = mul2(v.y, g.y); v.z = mul2(v.z, g.z); v.w = mul2(v.w, g.w);
        }
        U4 o;
        o.x = B::sel(sw, v.z, v.x); o.y = B::sel(sw, v.w, v.y);
        o.z = B::sel(sw, v.x, v.z); o.w = B::sel(sw, v.y, v.w);
        B::lds_w128(off + pl * GEO::PLANE, o, B::ptrue());
      }
    }
  }
  template <int NC = NCH>
  static FFC_FN void rows_in(const ConvArgs& a, int h, int pq, Unit un) {
    RowRegsT<NC> X;
    rows_load<NC>(a, h, pq, un, X);
    rows_store<NC>(a, h, pq, un, X);
  }
  // ------------------------------------------------------------------ input rows by LDS-DMA (round 4)
  // 32-point outer digit, L <= N/2 (HALF): E rows n1 >= 16 of a wave's column slice are dead from the moment its phase C has
  // read them until its next phase A writes them -- exactly the stretch in which the wave stores the current pair's output
  // rows.  The next pair's input rows are copied into those dead rows by `global_load_lds_dword` (no VGPR destination, no
  // LDS store pass, the copy runs under the output stores), in NATURAL layout (row 16 + n1, no bank swizzle: only phase A
  // reads them, 4 bytes per lane with an 8-byte lane stride, conflict-free as it is).  One instruction = 64 lanes x 4 bytes =
  // the wave's 128-column slice of one row of one plane: LDS-DMA destinations are lane-linear, and the slice is the largest
  // piece of E that is both contiguous and owned by one wave (16-byte pieces would cover four waves' slices and need barriers).
  // The backward kernels run on a 128-VGPR budget and cannot hold a register prefetch.
  // In-place safety of phase A: tile pair tp reads dword tp of the 8-byte chunks of rows 16.., and writes dword tp of (swizzled)
  // chunks of all rows: the dwords of the two tile pairs never meet.
  static constexpr bool HAS_DMA = GEO::OUTER && GEO::N1 == 32 && GEO::S1 == 1;
  static constexpr int DMA_ROW0 = GEO::N1 / 2;
  static FFC_FN void rows_dma(const ConvArgs& a, int h, int pq, Unit un) {
    if (a.stream) rows_dma_t<true>(a, h, pq, un);      // (one branch for the whole burst, not one per instruction)
    else rows_dma_t<false>(a, h, pq, un);
  }
  template <bool NT>
  static FFC_FN void rows_dma_t(const ConvArgs& a, int h, int pq, Unit un) {
    const i32 lane = B::opaque(B::lane());
    // Straight-line burst (round 4, late): no skip for rows beyond L or for the missing row of an odd batch -- they copy the clamped
    // last element(s) of a valid row and rows_dma_finish zeroes them like any tail.  With a wave-uniform `continue` per row every
    // copy sat in its own block, and in the prologue of the pair loop the compiler put a `s_waitcnt vmcnt(0)` in front of each
    // one (ISA of bwd_kernel<.., ZM = 1>: `D W0` 32 times = 32 memory round trips in a row at the start of every head).
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
      const int b = 2 * pq + pl;
      const uint16_t* base = (const uint16_t*)a.u + row_off(b < a.B ? b : a.B - 1, true, a.sbu, h, a.L);
#pragma unroll
      for (int r = 0; r < GEO::N1 / 2; r++) {
        i32 n = B::imin(lane * 2 + (r * GEO::Mi + un.wq * 128), a.L - 2);      // clamped: the tail is zeroed after the wait
        B::template g2lds32<NT>(base, n >> 1, un.eb + pl * GEO::PLANE + (DMA_ROW0 + r) * (GEO::Mi * 2) + un.wq * 256);
      }
    }
  }
  // wait for the wave's own copies (the reads that follow are its own: no barrier needed), zero what lies beyond L / B
  static FFC_FN void rows_dma_finish(const ConvArgs& a, int pq, Unit un) {
    B::vm_wait0();
    const i32 lane = B::opaque(B::lane());
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
      const bool ok = (2 * pq + pl) < a.B;
#pragma unroll
      for (int r = 0; r < GEO::N1 / 2; r++) {
        if (ok && (r + 1) * GEO::Mi <= a.L) continue;           // fully valid row (wave-uniform): nothing to do
        i32 n = lane * 2 + (r * GEO::Mi + un.wq * 128);
        B::lds_w32p(lane * 4 + (un.eb + pl * GEO::PLANE + (DMA_ROW0 + r) * (GEO::Mi * 2) + un.wq * 256), B::uconst(0),
                    B::pnot((n < a.L) && ok));
      }
    }
    B::lds_fence();
  }

  // inner-only sizes: per-lane batch row.  Element offset of (b,h,n) relative to tensor base fits
  // 32 bits in 16-byte units (launcher checks the tensor size).
  static FFC_FN U4 gload8_rows(const uint16_t* base, i32 b, int h, const ConvArgs& a, int64_t sb, i32 n, int fast, pred ok) {
    const int sbi = (int)sb;      // the launcher checks B * stride < 2^31 (and stride % 8 == 0 for the fast path)
    if (fast) {
      i32 o16 = b * (sbi >> 3) + (h * (a.L >> 3) + (n >> 3));
      return B::g_r128p(base, o16, ok && (n < a.L));
    }
    u32 w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      i32 e0 = b * sbi + (h * a.L + n + 2 * q);
      u32 lo = B::g_r16(base, e0, ok && ((n + 2 * q) < a.L));
      u32 hi = B::g_r16(base, e0 + 1, ok && ((n + (2 * q + 1)) < a.L));
      w[q] = lo | (hi << 16);
    }
    U4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
  }
  static FFC_FN void gstore8_rows(uint16_t* base, i32 b, int h, const ConvArgs& a, int64_t sb, i32 n, int fast, pred ok, U4 v) {
    const int sbi = (int)sb;
    if (fast) {
      i32 o16 = b * (sbi >> 3) + (h * (a.L >> 3) + (n >> 3));
      B::g_w128(base, o16, v, ok && (n < a.L));
      return;
    }
    u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      i32 e0 = b * sbi + (h * a.L + n + 2 * q);
      B::g_w16(base, e0, w[q] & 0xffffu, ok && ((n + 2 * q) < a.L));
      B::g_w16(base, e0 + 1, w[q] >> 16, ok && ((n + (2 * q + 1)) < a.L));
    }
  }

  // output rows with an output gate on the 16-byte path: the gate loads of a batch of chunks first (see rows_store_g)
#ifndef FFC_GATE_BATCH_OUT
#define FFC_GATE_BATCH_OUT FFC_GATE_BATCH      // (the output side holds no prefetched rows: the backward kernels afford a larger batch here)
#endif
  template <int NC>
  static FFC_FN void rows_out_g(const ConvArgs& a, int h, int pq, Unit un) {
    constexpr int GB = FFC_GATE_BATCH_OUT <= 0 ? 1 : (NC < FFC_GATE_BATCH_OUT ? NC : FFC_GATE_BATCH_OUT);
    static_assert(NC % GB == 0, "gate batch");
    const i32 lane = B::opaque(B::lane());
    const int fast = a.stream ? 2 : 1;
#pragma unroll
    for (int ib = 0; ib < NC; ib += GB) {
      U4 G[GB][2];
#pragma unroll
      for (int jj = 0; jj < GB; jj++) {
        const int i = ib + jj;
        i32 idx = lane + i * 64;
        i32 row = idx / CPR, m = (idx % CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
          if constexpr (GEO::OUTER) {
            const int b = 2 * pq + pl;
            const bool ok = b < a.B;
            G[jj][pl] = gload8((const uint16_t*)a.postgate + row_off(b, ok, a.sbp, h, a.L), row * GEO::Mi + m, a.L, fast, ok);
          } else {
            i32 b = (row + pq * GEO::G) * 2 + pl;
            G[jj][pl] = gload8_rows((const uint16_t*)a.postgate, b, h, a, a.sbp, m, fast, b < a.B);
          }
        }
      }
      B::sched_fence();
#pragma unroll
      for (int jj = 0; jj < GB; jj++) {
        const int i = ib + jj;
        i32 idx = lane + i * 64;
        i32 row = idx / CPR, m = (idx % CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
        pred sw;
        i32 off = pair_off(row, m, &sw) + un.eb;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
          U4 o = B::lds_r128(off + pl * GEO::PLANE);
          U4 v;
          v.x = B::sel(sw, o.z, o.x); v.y = B::sel(sw, o.w, o.y);
          v.z = B::sel(sw, o.x, o.z); v.w = B::sel(sw, o.y, o.w);
          v = mul4(v, G[jj][pl]);
          if constexpr (GEO::OUTER) {
            const int b = 2 * pq + pl;
            const bool ok = b < a.B;
            gstore8((uint16_t*)a.y + row_off(b, ok, a.sby, h, a.L), row * GEO::Mi + m, a.L, fast, ok, v);
          } else {
            i32 b = (row + pq * GEO::G) * 2 + pl;
            gstore8_rows((uint16_t*)a.y, b, h, a, a.sby, m, fast, b < a.B, v);
          }
        }
      }
    }
  }
  template <int NC = NCH>
  static FFC_FN void rows_out(const ConvArgs& a, int h, int pq, Unit un) {
    if constexpr (FFC_GATE_BATCH > 0) {
      if (a.fast && a.postgate) { rows_out_g<NC>(a, h, pq, un); return; }
    }
    const i32 lane = B::opaque(B::lane());
    const int fast = a.fast ? (a.stream ? 2 : 1) : 0;     // 2: streaming (non-temporal) fast path
#pragma unroll
    for (int i = 0; i < NC; i++) {
      i32 idx = lane + i * 64;
      i32 row = idx / CPR, m = (idx % CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
      pred sw;
      i32 off = pair_off(row, m, &sw) + un.eb;
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        U4 o = B::lds_r128(off + pl * GEO::PLANE);
        U4 v;
        v.x = B::sel(sw, o.z, o.x); v.y = B::sel(sw, o.w, o.y);
        v.z = B::sel(sw, o.x, o.z); v.w = B::sel(sw, o.y, o.w);
        if constexpr (GEO::OUTER) {
          const int b = 2 * pq + pl;
          const bool ok = b < a.B;
          i32 n = row * GEO::Mi + m;
          if (a.postgate) {
            U4 g = gload8((const uint16_t*)a.postgate + row_off(b, ok, a.sbp, h, a.L), n, a.L, fast, ok);
            v.x = mul2(v.x, g.x); v.y = mul2(v.y, g.y); v.z = mul2(v.z, g.z); v.w = mul2(v.w, g.w);
          }
          gstore8((uint16_t*)a.y + row_off(b, ok, a.sby, h, a.L), n, a.L, fast, ok, v);
        } else {
          i32 b = (row + pq * GEO::G) * 2 + pl;
          pred ok = b < a.B;
          if (a.postgate) {
            U4 g = gload8_rows((const uint16_t*)a.postgate, b, h, a, a.sbp, m, fast, ok);
            v.x = mul2(v.x, g.x); v.y = mul2(v.y, g.y); v.z = mul2(v.z, g.z); v.w = mul2(v.w, g.w);
          }
          gstore8_rows((uint16_t*)a.y, b, h, a, a.sby, m, fast, ok, v);
        }
      }
    }
  }

  // ------------------------------------------------------------------ multi-pass sizes: rows of pass k0 (see struct Pass)
  // a + sb * b on packed dtype pairs, fp32 arithmetic, one rounding
  static FFC_FN u32 add2(u32 a, u32 b, float sb) {
    f32 lo = B::template unpack_lo<DT>(a) + B::template unpack_lo<DT>(b) * sb;
    f32 hi = B::template unpack_hi<DT>(a) + B::template unpack_hi<DT>(b) * sb;
    return B::template pack<DT>(lo, hi);
  }
  static FFC_FN U4 add4(const U4& a, const U4& b, float sb) {
    U4 o; o.x = add2(a.x, b.x, sb); o.y = add2(a.y, b.y, sb); o.z = add2(a.z, b.z, sb); o.w = add2(a.w, b.w, sb);
    return o;
  }
  static FFC_FN U4 mul4(const U4& v, const U4& g) {
    U4 o; o.x = mul2(v.x, g.x); o.y = mul2(v.y, g.y); o.z = mul2(v.z, g.z); o.w = mul2(v.w, g.w);
    return o;
  }
  static FFC_FN U4 mask4(const U4& v, pred ok) {
    U4 o; o.x = B::sel(ok, v.x, B::uconst(0)); o.y = B::sel(ok, v.y, B::uconst(0));
    o.z = B::sel(ok, v.z, B::uconst(0)); o.w = B::sel(ok, v.w, B::uconst(0));
    return o;
  }
  // (u * pregate)[n .. n+7] of batch row b, zero beyond L / for a missing row
  static FFC_FN U4 load_gated(const ConvArgs& a, int h, int b, i32 n, int fast) {
    const bool ok = b < a.B;
    U4 v = gload8((const uint16_t*)a.u + row_off(b, ok, a.sbu, h, a.L), n, a.L, fast, ok);
    if (fast) v = mask4(v, (n < a.L) && ok);
    if (a.pregate) v = mul4(v, gload8((const uint16_t*)a.pregate + row_off(b, ok, a.sbg, h, a.L), n, a.L, fast, ok));
    return v;
  }
  // rows_in of pass k0: E row n1 = sum_n0 W_R^{n0 k0} (u * pregate)[n0 M + n1 Mi + m]; plane 0 = Re (batch row 2p),
  // plane 1 = Im (row 2p+1).  W_R^{q'} = (-i)^q with q = q' * 4 / R:  (-i)^q (r + i s) = (r,s), (s,-r), (-r,-s), (-s,r).
  // (round 4: the loads of a batch of chunks -- rows and, gated, their gates -- are issued together; before, each chunk's two loads
  // sat between the previous chunk's LDS writes and an `unroll 1` loop the scheduler cannot move loads across: 8 round trips per pass)
  static constexpr int GB_RP = FFC_GATE_BATCH <= 0 ? 1 : FFC_GATE_BATCH;
  // The 16-byte / element-wise decision of the multi-pass row functions is made ONCE per call (rows_*_rp_t<NC, FASTP>), not inside every
  // gload8 (round 4, last change of the round).  With the three-way switch inside, each load of a batch sat in its own flow block whose
  // merge carries a `s_waitcnt vmcnt(0)` -- load, wait, load, wait ..., 16 round trips per pass and wave in the UNGATED kernels too (ISA
  // of conv_rp_kernel / conv_kernel<Geo<1,32,32>>); rows_store_g / rows_out_g pass a constant 1 / 2 and never had this.  Same box, bit-
  // identical (profiles/r04_ab_rp_hoist.txt): fft 65536 B16 H768 forward 1.576 -> 1.241 ms, training forward 1.730 -> 1.383; fft 131072
  // B8 H768 2.338 -> 1.827 / 2.536 -> 1.951.  Forward kernels only: in the multi-pass BACKWARD kernels the second copy of the row code
  // overflows the 128-VGPR budget (build audit) -- next: the fast / element-wise split as two kernels (FFC_RP_HOIST=0: the former code).
#ifndef FFC_RP_HOIST
#define FFC_RP_HOIST 1
#endif
// chunks per batch of the multi-pass row functions in the fast-only BACKWARD kernels (128-VGPR budget; the forward kernels take 4)
#ifndef FFC_RP_BATCH_LEAN
#define FFC_RP_BATCH_LEAN 2
#endif
// 0: the multi-pass backward always runs the kernel with the run-time access-width switch (A/B builds, bit-identity test)
#ifndef FFC_RP_FASTK
#define FFC_RP_FASTK 1
#endif
  template <int NC>
  static FFC_FN void rows_in_rp(const ConvArgs& a, int h, int pq, Unit un, Pass ps) {
    if constexpr (B::FAST_ONLY) {
      // round 5, multi-pass BACKWARD kernels (the 16-byte path as a kernel instantiation of its own, bwd_rp_kernel<.., FASTK = true>):
      // rows that fit one block (L <= M: the padded case) are the single-pass row load -- sum_n0 has one term, the pass factor lives
      // in the outer-digit matrix and the twiddle phase -- with every load of the wave's slice in flight; longer rows in batches
      if (a.L <= GEO::N) { rows_in<NC>(a, h, pq, un); return; }
      rows_in_rp_t<NC, true>(a, h, pq, un, ps);
      return;
    }
    if constexpr (FFC_RP_HOIST != 0 && !B::LEAN_OUTER) {
      if (a.fast) { rows_in_rp_t<NC, true>(a, h, pq, un, ps); return; }
    }
    rows_in_rp_t<NC, false>(a, h, pq, un, ps);
  }
  template <int NC, bool FASTP>
  static FFC_FN void rows_in_rp_t(const ConvArgs& a, int h, int pq, Unit un, Pass ps) {
    constexpr int GB0 = (B::FAST_ONLY && B::LEAN_OUTER) ? FFC_RP_BATCH_LEAN : GB_RP;
    constexpr int GB = NC < GB0 ? NC : GB0;
    static_assert(NC % GB == 0, "row batch");
    const i32 lane = B::opaque(B::lane());
    const int fast = FASTP ? (a.stream ? 2 : 1) : (a.fast ? (a.stream ? 2 : 1) : 0);
    const int n0max = (a.L + GEO::N - 1) / GEO::N;
    const bool hasg = a.pregate != nullptr;
    int64_t ru[2], rg[2]; bool okb[2];
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
      okb[pl] = (2 * pq + pl) < a.B;
      ru[pl] = row_off(2 * pq + pl, okb[pl], a.sbu, h, a.L);
      rg[pl] = row_off(2 * pq + pl, okb[pl], a.sbg, h, a.L);
    }
#pragma unroll
    for (int ib = 0; ib < NC; ib += GB) {
      U4 acc[GB][2];
#pragma unroll 1
      for (int n0 = 0; n0 < n0max; n0++) {
        U4 W[GB][2], G[GB][2];
#pragma unroll
        for (int jj = 0; jj < GB; jj++) {
          i32 idx = lane + (ib + jj) * 64;
          i32 n = (idx / CPR) * GEO::Mi + (idx % CPR) * 8 + un.wq * 128 * GEO::S1 + n0 * GEO::N;
#pragma unroll
          for (int pl = 0; pl < 2; pl++) {
            W[jj][pl] = gload8((const uint16_t*)a.u + ru[pl], n, a.L, fast, okb[pl]);
            if (hasg) G[jj][pl] = gload8((const uint16_t*)a.pregate + rg[pl], n, a.L, fast, okb[pl]);
          }
        }
        B::sched_fence();
        const int q = (n0 * ps.k0 * (4 / ps.R)) & 3;
        const float sgr = q < 2 ? 1.0f : -1.0f, sgi = (q == 0 || q == 3) ? 1.0f : -1.0f;
#pragma unroll
        for (int jj = 0; jj < GB; jj++) {
          i32 idx = lane + (ib + jj) * 64;
          i32 n = (idx / CPR) * GEO::Mi + (idx % CPR) * 8 + un.wq * 128 * GEO::S1 + n0 * GEO::N;
          U4 w[2];
#pragma unroll
          for (int pl = 0; pl < 2; pl++) {      // (u * pregate)[n .. n+7], zero beyond L / for a missing row (load_gated)
            w[pl] = W[jj][pl];
            if (fast) w[pl] = mask4(w[pl], (n < a.L) && okb[pl]);
            if (hasg) w[pl] = mul4(w[pl], G[jj][pl]);
          }
          if (n0 == 0) { acc[jj][0] = w[0]; acc[jj][1] = w[1]; }
          else if (q & 1) { acc[jj][0] = add4(acc[jj][0], w[1], sgr); acc[jj][1] = add4(acc[jj][1], w[0], sgi); }
          else { acc[jj][0] = add4(acc[jj][0], w[0], sgr); acc[jj][1] = add4(acc[jj][1], w[1], sgi); }
        }
      }
#pragma unroll
      for (int jj = 0; jj < GB; jj++) {
        i32 idx = lane + (ib + jj) * 64;
        i32 row = idx / CPR, m = (idx % CPR) * 8 + un.wq * 128 * GEO::S1;
        pred sw;
        i32 off = pair_off(row, m, &sw) + un.eb;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
          const U4& v = acc[jj][pl];
          U4 o;
          o.x = B::sel(sw, v.z, v.x); o.y = B::sel(sw, v.w, v.y);
          o.z = B::sel(sw, v.x, v.z); o.w = B::sel(sw, v.y, v.w);
          B::lds_w128(off + pl * GEO::PLANE, o, B::ptrue());
        }
      }
    }
  }
  // Multi-pass sizes: the side product of the row load (ConvArgs::aux_in, see rows_store) as a pass of its own over the wave's
  // column slice, run once per pair (pass 0) ahead of rows_in_rp, which then finds the rows of `u` in L2.  (Folded into
  // load_gated it overflowed the 128-VGPR budget of the multi-pass backward kernel; build.py check_agpr.)
  template <int NC>
  static FFC_FN void rows_aux_rp(const ConvArgs& a, int h, int pq, Unit un) {
    const i32 lane = B::opaque(B::lane());
    const int fast = a.fast ? (a.stream ? 2 : 1) : 0;
    const int n0max = (a.L + GEO::N - 1) / GEO::N;
#pragma unroll 1
    for (int n0 = 0; n0 < n0max; n0++) {
#pragma unroll 1
      for (int i = 0; i < NC; i++) {
        i32 idx = lane + i * 64;
        i32 n = (idx / CPR) * GEO::Mi + (idx % CPR) * 8 + un.wq * 128 * GEO::S1 + n0 * GEO::N;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
          const int b = 2 * pq + pl;
          const bool ok = b < a.B;
          U4 v = gload8((const uint16_t*)a.u + row_off(b, ok, a.sbu, h, a.L), n, a.L, fast, ok);
          U4 r = gload8((const uint16_t*)a.aux_in + row_off(b, ok, a.sbai, h, a.L), n, a.L, fast, ok);
          gstore8((uint16_t*)a.aux_out + row_off(b, ok, a.sbao, h, a.L), n, a.L, fast, ok, mul4(v, r));
        }
      }
    }
  }
  // rows_out of pass k0: y[n0 M + m] (+)= i^q y_k0[m], the last pass (* postgate), q = n0 k0 4/R: i^q (r + i s) = (r,s), (-s,r), (-r,-s),
  // (s,-r).  Passes k0 > 0 add to what the SAME wave stored in the earlier passes (its own column slice).
  template <int NC>
  static FFC_FN void rows_out_rp(const ConvArgs& a, int h, int pq, Unit un, Pass ps) {
    if constexpr (B::FAST_ONLY) { rows_out_rp_t<NC, true>(a, h, pq, un, ps); return; }
    if constexpr (FFC_RP_HOIST != 0 && !B::LEAN_OUTER) {
      if (a.fast) { rows_out_rp_t<NC, true>(a, h, pq, un, ps); return; }
    }
    rows_out_rp_t<NC, false>(a, h, pq, un, ps);
  }
  template <int NC, bool FASTP>
  static FFC_FN void rows_out_rp_t(const ConvArgs& a, int h, int pq, Unit un, Pass ps) {
    // batches of chunks: the earlier passes' sums (passes k0 > 0) and, on the last pass, the output gate of a batch are requested
    // together ahead of the work on them (round 4; before: the sums up front in the forward kernels only, the gate chunk by chunk)
    constexpr int GBL = B::FAST_ONLY ? FFC_RP_BATCH_LEAN : GB_RP;
    constexpr int GB = B::LEAN_OUTER ? (NC < GBL ? NC : GBL) : (NC < 4 ? NC : 4);
    static_assert(NC % GB == 0, "row batch");
    const i32 lane = B::opaque(B::lane());
    const int fast = FASTP ? (a.stream ? 2 : 1) : (a.fast ? (a.stream ? 2 : 1) : 0);
    const int n0max = (a.L + GEO::N - 1) / GEO::N;
    const bool add_old = ps.k0 > 0, gate = a.postgate && ps.k0 == ps.R - 1;
    int64_t ro[2], rg[2]; bool okb[2];
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
      okb[pl] = (2 * pq + pl) < a.B;
      ro[pl] = row_off(2 * pq + pl, okb[pl], a.sby, h, a.L);
      rg[pl] = row_off(2 * pq + pl, okb[pl], a.sbp, h, a.L);
    }
#pragma unroll 1
    for (int n0 = 0; n0 < n0max; n0++) {
      const int q = (n0 * ps.k0 * (4 / ps.R)) & 3;
#pragma unroll
      for (int ib = 0; ib < NC; ib += GB) {
        U4 old[GB][2], G[GB][2];
#pragma unroll
        for (int jj = 0; jj < GB; jj++) {
          i32 idx = lane + (ib + jj) * 64;
          i32 n = (idx / CPR) * GEO::Mi + (idx % CPR) * 8 + un.wq * 128 * GEO::S1 + n0 * GEO::N;
#pragma unroll
          for (int pl = 0; pl < 2; pl++) {
            if (add_old) old[jj][pl] = gload8((const uint16_t*)a.y + ro[pl], n, a.L, fast, okb[pl]);
            if (gate) G[jj][pl] = gload8((const uint16_t*)a.postgate + rg[pl], n, a.L, fast, okb[pl]);
          }
        }
        B::sched_fence();
#pragma unroll
        for (int jj = 0; jj < GB; jj++) {
          i32 idx = lane + (ib + jj) * 64;
          i32 row = idx / CPR, m = (idx % CPR) * 8 + un.wq * 128 * GEO::S1;
          pred sw;
          i32 off = pair_off(row, m, &sw) + un.eb;
          U4 y[2];
#pragma unroll
          for (int pl = 0; pl < 2; pl++) {
            U4 o = B::lds_r128(off + pl * GEO::PLANE);
            y[pl].x = B::sel(sw, o.z, o.x); y[pl].y = B::sel(sw, o.w, o.y);
            y[pl].z = B::sel(sw, o.x, o.z); y[pl].w = B::sel(sw, o.y, o.w);
          }
          i32 n = row * GEO::Mi + m + n0 * GEO::N;
#pragma unroll
          for (int pl = 0; pl < 2; pl++) {
            // plane 0: {r, -s, -r, s}[q], plane 1: {s, r, -s, -r}[q]
            U4 c = ((q & 1) != 0) == (pl == 0) ? y[1] : y[0];
            const float sg = pl == 0 ? ((q == 0 || q == 3) ? 1.0f : -1.0f) : (q < 2 ? 1.0f : -1.0f);
            // the passes' contributions are summed ungated; the output gate multiplies the sum, on the last pass only (the gate
            // load and its 28 VALU per 8 elements were 20 % of a pass's VALU count when every pass multiplied its own part)
            if (add_old) c = add4(old[jj][pl], c, sg);
            if (gate) c = mul4(c, G[jj][pl]);
            gstore8((uint16_t*)a.y + ro[pl], n, a.L, fast, okb[pl], c);
          }
        }
      }
    }
  }

  // ------------------------------------------------------------------ phases A / C (outer DFT, in place)
  // HALF (L <= N/2): E rows n1 >= N1/2 carry no input and their outputs lie beyond L.  c = tile-local row constant
  // (bit 2 clear; the lane adds 4*hi): the row is dead iff (c mod N1) >= N1/2.
  static constexpr bool row_dead(int c) { return (c % GEO::N1) >= GEO::N1 / 2; }

  // The wave owns columns [wq*128*S1, +128*S1) of every E row: 4 tiles t, lane j <-> column
  // s1*128 + 4j + t.  FWD: rows are n1 (real pair x), result rows k1 with the W_N^{m k1} twiddle.
  // !FWD: rows are k1, result rows n1 (re -> batch row 2p, im -> row 2p+1).
  // HALF: input rows n1 >= 16 are all zero (L <= 16*Mi, 32-point outer digit) -> one K-step.
  template <bool FWD, bool HALF, bool RP = false, bool NOTW = false>
  static FFC_FN void outer_stage_tile(int L, Unit un, float s_fwd = 1.0f, Pass ps = Pass()) {
    const i32 lane = B::opaque(B::lane());
    const int w = un.wq;
    const i32 j = lane & 31, hi = lane >> 5;
    constexpr int ms_lim = (FWD && HALF && GEO::N1 == 32) ? 1 : 2;      // a whole K-step of dead rows (32-point digit)
    // tile-local row R = 4*hi + c (c a compile-time constant with bit 2 clear) -> column set
    // s1 = c / N1 and E row rw = c % N1 + 4*hi.  e_off = row term + swizzled column term, so every
    // access below is (one of S1 lane-dependent bases) + immediate.
    i32 colb[GEO::S1];
#pragma unroll
    for (int s = 0; s < GEO::S1; s++)
      colb[s] = e_off<GEO, i32>(hi * 4, j * 4 + (s * 128 + w * 128 * GEO::S1)) + un.eb;
    Mat F1;
    if constexpr (RP) load_mat(F1, FWD ? ps.mat_fwd : ps.mat_inv, lane);   // this pass's table (6 KB, L2-resident)
    else lds_mat(F1, GEO::L_F1);
#pragma unroll 1
    for (int t = 0; t < 4; t++) {          // the 4 column tiles of this wave; a runtime loop bounds the live ranges
      // operand halves come straight from 16-bit LDS reads (element t of each 8-byte chunk) and the results go
      // back as 16-bit stores: no word stash across tiles, no shift/mask work to split or merge dwords
      i32 colt[GEO::S1];
#pragma unroll
      for (int s = 0; s < GEO::S1; s++) colt[s] = colb[s] + 2 * t;
      Op op;
#pragma unroll
      for (int ms = 0; ms < 2; ms++) {
        if (ms >= ms_lim) continue;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 vr[2], vi[2];
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {
            const int e = 2 * d + hf;
            const int c = 16 * ms + 8 * (e >> 2) + (e & 3);
            const int s1 = c / GEO::N1, rwc = c % GEO::N1;
            i32 off = colt[s1] + rwc * (GEO::Mi * 2);
            // no length mask: rows_store zero-fills every E row this stage reads (HALF never reads the dead rows)
            if (FWD && HALF && row_dead(c)) { vr[hf] = B::uconst(0); vi[hf] = B::uconst(0); continue; }
            vr[hf] = B::lds_r16(off); vi[hf] = B::lds_r16(off + GEO::PLANE);
          }
          op.r[ms][d] = vr[0] | (vr[1] << 16);
          op.i[ms][d] = vi[0] | (vi[1] << 16);
        }
      }
      A16 re, im;
      re = B::a16_zero(); im = B::a16_zero();
      cmm<!FWD, false>(re, im, op, F1, ms_lim);
      if (FWD && !NOTW) {
        // s_fwd * W_N^{m*k1}: registers <-> rows k1 = 4*hi + {0..3} + 8*{0..3} (mod N1), lane/tile <-> column m
        if constexpr (CHAIN16_A && FFC_CHAIN16_TILE) {              // both halves share the column m: one chain (twiddle16)
          i32 m = j * 4 + (w * 128 * GEO::S1 + t);
          i32 k0 = hi * 4;
          if constexpr (RP) twiddle16(re, im, m * (k0 * ps.R + ps.k0), m * ps.R, -1.0f, s_fwd, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
          else twiddle16(re, im, m * k0, m, -1.0f, s_fwd);
        } else
#pragma unroll
        for (int half = 0; half < 2; half++) {     // accumulator registers 0-7 / 8-15 (rows +16)
          const int s1 = (16 * half) / GEO::N1;     // second half: next column set when N1 == 16
          i32 m = j * 4 + (s1 * 128 + w * 128 * GEO::S1 + t);
          i32 k0 = hi * 4 + ((16 * half) % GEO::N1);
          F2 tr[4], ti[4];
          if constexpr (RP) chain8p(m * (k0 * ps.R + ps.k0), m * ps.R, -1.0f, s_fwd, tr, ti, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
          else chain8p(m * k0, m, -1.0f, s_fwd, tr, ti);
          apply8(re, im, half, tr, ti);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        if (!FWD && HALF && row_dead((r & 3) + 8 * (r >> 2))) continue;   // rows beyond L: never stored
        u32 vr = B::template pack<DT>(re[r], re[r + 1]);
        u32 vi = B::template pack<DT>(im[r], im[r + 1]);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int c = ((r + q) & 3) + 8 * ((r + q) >> 2);
          const int s1 = c / GEO::N1, rwc = c % GEO::N1;
          i32 off = colt[s1] + rwc * (GEO::Mi * 2);
          B::lds_w16(off, q ? (vr >> 16) : vr);
          B::lds_w16(off + GEO::PLANE, q ? (vi >> 16) : vi);
        }
      }
    }
  }

  // tile-pair variant: 32-bit LDS accesses shared by two column tiles.  Dword tp of each 8-byte chunk holds elements
  // (2tp, 2tp+1) of a row = the same row of the two tiles: operand dwords are one 16-bit-half merge each
  // (B::merge_lo / merge_hi = v_perm_b32), and the results go back as one v_cvt_pk(tile 2tp, tile 2tp+1) per row and plane
  // (the first tile's fp32 accumulators wait for the second tile's instead of a stash of packed halves).
  // No length masks: every E row this stage reads was written by rows_store, zero beyond L (HALF never reads rows >= 16).
  // DIN: the input rows were copied by LDS-DMA into E rows 16.. in natural layout (rows_dma)
  template <bool FWD, bool HALF, bool RP = false, bool DIN = false, bool NOTW = false>
  static FFC_FN void outer_stage_pair(int L, Unit un, float s_fwd = 1.0f, Pass ps = Pass()) {
    static_assert(!DIN || (FWD && HALF && HAS_DMA && !RP), "DMA input rows: forward stage of the half-empty 32-point digit");
    const i32 lane = B::opaque(B::lane());
    const int w = un.wq;
    const i32 j = lane & 31, hi = lane >> 5;
    constexpr int ms_lim = (FWD && HALF && GEO::N1 == 32) ? 1 : 2;      // a whole K-step of dead rows (32-point digit)
    i32 colr = hi * (4 * GEO::Mi * 2) + j * 8 + (un.eb + DMA_ROW0 * (GEO::Mi * 2) + w * 256);     // DIN: natural-layout read base
    // tile-local row R = 4*hi + c (c a compile-time constant with bit 2 clear) -> column set
    // s1 = c / N1 and E row rw = c % N1 + 4*hi.  e_off = row term + swizzled column term, so every
    // access below is (one of S1 lane-dependent bases) + immediate.
    i32 colb[GEO::S1];
#pragma unroll
    for (int s = 0; s < GEO::S1; s++)
      colb[s] = e_off<GEO, i32>(hi * 4, j * 4 + (s * 128 + w * 128 * GEO::S1)) + un.eb;
    Mat F1;
    if constexpr (RP) load_mat(F1, FWD ? ps.mat_fwd : ps.mat_inv, lane);   // this pass's table (6 KB, L2-resident)
    else lds_mat(F1, GEO::L_F1);
#pragma unroll 1
    for (int tp = 0; tp < 2; tp++) {       // tiles (2tp, 2tp+1): a runtime loop bounds the live ranges
      u32 rawr[2][8], rawi[2][8];
#pragma unroll
      for (int ms = 0; ms < 2; ms++) {
        if (ms >= ms_lim) continue;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int c = 16 * ms + 8 * (e >> 2) + (e & 3);
          const int s1 = c / GEO::N1, rwc = c % GEO::N1;
          i32 off = (DIN ? colr : colb[s1]) + (rwc * (GEO::Mi * 2) + 4 * tp);
          if (FWD && HALF && row_dead(c)) { rawr[ms][e] = B::uconst(0); rawi[ms][e] = B::uconst(0); continue; }
          rawr[ms][e] = B::lds_r32(off); rawi[ms][e] = B::lds_r32(off + GEO::PLANE);
        }
      }
      A16 re0, im0;
      u32 s0r[8], s0i[8];
#pragma unroll
      for (int th = 0; th < 2; th++) {
        Op op;
#pragma unroll
        for (int ms = 0; ms < 2; ms++) {
          if (ms >= ms_lim) continue;
#pragma unroll
          for (int d = 0; d < 4; d++) {
            op.r[ms][d] = th ? B::merge_hi(rawr[ms][2 * d], rawr[ms][2 * d + 1]) : B::merge_lo(rawr[ms][2 * d], rawr[ms][2 * d + 1]);
            op.i[ms][d] = th ? B::merge_hi(rawi[ms][2 * d], rawi[ms][2 * d + 1]) : B::merge_lo(rawi[ms][2 * d], rawi[ms][2 * d + 1]);
          }
        }
        A16 re, im;
        re = B::a16_zero(); im = B::a16_zero();
        cmm<!FWD, false>(re, im, op, F1, ms_lim);
        if (FWD && !NOTW) {
          // s_fwd * W_N^{m*k1}: registers <-> rows k1 = 4*hi + {0..3} + 8*{0..3} (mod N1), lane/tile <-> column m
          if constexpr (CHAIN16_A) {              // both halves share the column m: one chain (twiddle16)
            i32 m = j * 4 + (w * 128 * GEO::S1 + 2 * tp + th);
            i32 k0 = hi * 4;
            if constexpr (RP) twiddle16(re, im, m * (k0 * ps.R + ps.k0), m * ps.R, -1.0f, s_fwd, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
            else twiddle16(re, im, m * k0, m, -1.0f, s_fwd);
          } else
#pragma unroll
          for (int half = 0; half < 2; half++) {     // accumulator registers 0-7 / 8-15 (rows +16)
            const int s1 = (16 * half) / GEO::N1;     // second half: next column set when N1 == 16
            i32 m = j * 4 + (s1 * 128 + w * 128 * GEO::S1 + 2 * tp + th);
            i32 k0 = hi * 4 + ((16 * half) % GEO::N1);
            F2 tr[4], ti[4];
            if constexpr (RP) chain8p(m * (k0 * ps.R + ps.k0), m * ps.R, -1.0f, s_fwd, tr, ti, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
            else chain8p(m * k0, m, -1.0f, s_fwd, tr, ti);
            apply8(re, im, half, tr, ti);
          }
        }
        if constexpr (!B::LEAN_OUTER) {
          if (th == 0) {
            re0 = re; im0 = im;
          } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
              if (!FWD && HALF && row_dead((r & 3) + 8 * (r >> 2))) continue;   // rows beyond L: never stored
              const int c = (r & 3) + 8 * (r >> 2);
              const int s1 = c / GEO::N1, rwc = c % GEO::N1;
              i32 off = colb[s1] + (rwc * (GEO::Mi * 2) + 4 * tp);
              B::lds_w32(off, B::template pack<DT>(re0[r], re[r]));
              B::lds_w32(off + GEO::PLANE, B::template pack<DT>(im0[r], im[r]));
            }
          }
        } else {
          // backward kernels (128-VGPR budget): the first tile waits as 16 packed row pairs instead of 32 fp32 values;
          // the store words (tile 2tp | tile 2tp+1 of one row) are the 16-bit half merges of the two tiles' row pairs
          if (th == 0) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
              if (!FWD && HALF && row_dead(((2 * q) & 3) + 8 * ((2 * q) >> 2))) continue;
              s0r[q] = B::template pack<DT>(re[2 * q], re[2 * q + 1]);
              s0i[q] = B::template pack<DT>(im[2 * q], im[2 * q + 1]);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 8; q++) {
              if (!FWD && HALF && row_dead(((2 * q) & 3) + 8 * ((2 * q) >> 2))) continue;
              u32 p1r = B::template pack<DT>(re[2 * q], re[2 * q + 1]), p1i = B::template pack<DT>(im[2 * q], im[2 * q + 1]);
#pragma unroll
              for (int o = 0; o < 2; o++) {
                const int r = 2 * q + o;
                const int c = (r & 3) + 8 * (r >> 2);
                const int s1 = c / GEO::N1, rwc = c % GEO::N1;
                i32 off = colb[s1] + (rwc * (GEO::Mi * 2) + 4 * tp);
                B::lds_w32(off, o ? B::merge_hi(s0r[q], p1r) : B::merge_lo(s0r[q], p1r));
                B::lds_w32(off + GEO::PLANE, o ? B::merge_hi(s0i[q], p1i) : B::merge_lo(s0i[q], p1i));
              }
            }
          }
        }
      }
    }
  }

  // Round 5: all four column tiles of the wave in one pass (forward / dx kernels: 256-VGPR budget).  The tile-pair form reads and
  // writes 4 bytes per lane at an 8-byte lane stride: `ds_read_b32` / `ds_write_b32` bank addresses are (a / 4) mod 32 over 32-lane
  // groups (MI355X_MICROARCH.md, LDS), so lanes j and j + 16 meet on one bank -- every access of phases A / C was 2-way
  // conflicted (PMC, round 4: bank-conflict cycles = 33 % of the LDS-active cycles).  Here a lane moves the whole 8-byte chunk
  // (elements of tiles 0..3 of one row): `ds_read_b64` is conflict-free at that stride (32 lanes = one 256-byte bank row) and
  // `ds_write_b64` covers 16 lanes x 8 bytes per group; half the LDS instructions, a quarter of the LDS-array cycles.  The first
  // tile pair's results wait as packed row pairs (32 dwords) for the second pair's.  Same arithmetic as outer_stage_pair, so the
  // results are bit-identical (simulator test).  FFC_OUTER_QUAD=0: the tile-pair form (A/B builds).
#ifndef FFC_OUTER_QUAD
#define FFC_OUTER_QUAD 1
#endif
  template <bool FWD, bool HALF, bool RP = false, bool NOTW = false>
  static FFC_FN void outer_stage_quad(int L, Unit un, float s_fwd = 1.0f, Pass ps = Pass()) {
    const i32 lane = B::opaque(B::lane());
    const int w = un.wq;
    const i32 j = lane & 31, hi = lane >> 5;
    constexpr int ms_lim = (FWD && HALF && GEO::N1 == 32) ? 1 : 2;
    i32 colb[GEO::S1];
#pragma unroll
    for (int s = 0; s < GEO::S1; s++)
      colb[s] = e_off<GEO, i32>(hi * 4, j * 4 + (s * 128 + w * 128 * GEO::S1)) + un.eb;
    Mat F1;
    if constexpr (RP) load_mat(F1, FWD ? ps.mat_fwd : ps.mat_inv, lane);
    else lds_mat(F1, GEO::L_F1);
    U2 rawr[2][8], rawi[2][8];
#pragma unroll
    for (int ms = 0; ms < 2; ms++) {
      if (ms >= ms_lim) continue;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int c = 16 * ms + 8 * (e >> 2) + (e & 3);
        const int s1 = c / GEO::N1, rwc = c % GEO::N1;
        i32 off = colb[s1] + rwc * (GEO::Mi * 2);
        if (FWD && HALF && row_dead(c)) { rawr[ms][e].x = B::uconst(0); rawr[ms][e].y = B::uconst(0); rawi[ms][e] = rawr[ms][e]; continue; }
        rawr[ms][e] = B::lds_r64(off); rawi[ms][e] = B::lds_r64(off + GEO::PLANE);
      }
    }
    u32 p0r[16], p0i[16];          // tiles (0, 1) of every result row, packed, until tiles (2, 3) are done
#pragma unroll
    for (int tp = 0; tp < 2; tp++) {
      A16 re0, im0;
#pragma unroll
      for (int th = 0; th < 2; th++) {
        Op op;
#pragma unroll
        for (int ms = 0; ms < 2; ms++) {
          if (ms >= ms_lim) continue;
#pragma unroll
          for (int d = 0; d < 4; d++) {
            const u32 ar = tp ? rawr[ms][2 * d].y : rawr[ms][2 * d].x, br = tp ? rawr[ms][2 * d + 1].y : rawr[ms][2 * d + 1].x;
            const u32 ai = tp ? rawi[ms][2 * d].y : rawi[ms][2 * d].x, bi = tp ? rawi[ms][2 * d + 1].y : rawi[ms][2 * d + 1].x;
            op.r[ms][d] = th ? B::merge_hi(ar, br) : B::merge_lo(ar, br);
            op.i[ms][d] = th ? B::merge_hi(ai, bi) : B::merge_lo(ai, bi);
          }
        }
        A16 re, im;
        re = B::a16_zero(); im = B::a16_zero();
        cmm<!FWD, false>(re, im, op, F1, ms_lim);
        if (FWD && !NOTW) {
          if constexpr (CHAIN16_A) {
            i32 m = j * 4 + (w * 128 * GEO::S1 + 2 * tp + th);
            i32 k0 = hi * 4;
            if constexpr (RP) twiddle16(re, im, m * (k0 * ps.R + ps.k0), m * ps.R, -1.0f, s_fwd, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
            else twiddle16(re, im, m * k0, m, -1.0f, s_fwd);
          } else
#pragma unroll
          for (int half = 0; half < 2; half++) {
            const int s1 = (16 * half) / GEO::N1;
            i32 m = j * 4 + (s1 * 128 + w * 128 * GEO::S1 + 2 * tp + th);
            i32 k0 = hi * 4 + ((16 * half) % GEO::N1);
            F2 tr[4], ti[4];
            if constexpr (RP) chain8p(m * (k0 * ps.R + ps.k0), m * ps.R, -1.0f, s_fwd, tr, ti, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
            else chain8p(m * k0, m, -1.0f, s_fwd, tr, ti);
            apply8(re, im, half, tr, ti);
          }
        }
        if (th == 0) {
          re0 = re; im0 = im;
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            if (!FWD && HALF && row_dead((r & 3) + 8 * (r >> 2))) continue;   // rows beyond L: never stored
            const u32 vr = B::template pack<DT>(re0[r], re[r]), vi = B::template pack<DT>(im0[r], im[r]);
            if (tp == 0) {
              p0r[r] = vr; p0i[r] = vi;
            } else {
              const int c = (r & 3) + 8 * (r >> 2);
              const int s1 = c / GEO::N1, rwc = c % GEO::N1;
              i32 off = colb[s1] + rwc * (GEO::Mi * 2);
              U2 wr, wi;
              wr.x = p0r[r]; wr.y = vr; wi.x = p0i[r]; wi.y = vi;
              B::lds_w64(off, wr);
              B::lds_w64(off + GEO::PLANE, wi);
            }
          }
        }
      }
    }
  }

  // The forward/dx kernels take the tile-pair variant (fastest); the backward kernels, which run on the
  // architectural half of the register file, take the per-tile one (B::LEAN_OUTER).
  // NOTW: no outer twiddle behind the forward DFT (FFC_FOLD_TW: it is folded into the inner stages' per-tile matrices, tile_fwd<.., FOLD>)
  template <bool FWD, bool HALF, bool RP = false, bool DIN = false, bool NOTW = false>
  static FFC_FN void outer_stage(int L, Unit un, float s_fwd = 1.0f, Pass ps = Pass()) {
    if constexpr (DIN) { outer_stage_pair<FWD, HALF, RP, true, NOTW>(L, un, s_fwd, ps); return; }
#if defined(FFC_KO) && (FFC_KO & 8)
    return;
#endif
#ifndef FFC_LEAN_TILE
#define FFC_LEAN_TILE 0
#endif
    // backward kernels (128-VGPR budget): the tile-pair form only fits with one K-step of raw rows (half-empty outer digit)
    if constexpr (B::LEAN_OUTER && (FFC_LEAN_TILE || !(FWD && HALF && GEO::N1 == 32))) outer_stage_tile<FWD, HALF, RP, NOTW>(L, un, s_fwd, ps);
    // (HALF kernels only: same box, round 5 -- forward -1 ... -3 % at L <= N/2 (config 2 0.4474 -> 0.4334 ms with the chain change, gated
    // fft 16384 -2.8 %), but the full-length spectrum-saving forward came out 5 % SLOWER with it (0.695 -> 0.730 ms, no spills: 28 more
    // live registers through phase C), the full-length plain forward 1.5 % faster: profiles/r05_ab_kernels.txt)
    else if constexpr (!B::LEAN_OUTER && FFC_OUTER_QUAD != 0 && HALF) outer_stage_quad<FWD, HALF, RP, NOTW>(L, un, s_fwd, ps);
    else outer_stage_pair<FWD, HALF, RP, false, NOTW>(L, un, s_fwd, ps);
  }

  // ------------------------------------------------------------------ phase B (inner tile)
  // per-lane LDS offsets of tile 0 (tile tau adds tau*G rows): operand reads [K-step][rho] and write-back [rq]
  struct InnerRegs { Mat F2; CT16 tw; i32 roff[2][2]; i32 woff[4]; };
  template <bool TWR = true>
  static FFC_FN void load_inner(InnerRegs& R, Unit un) {
    lds_mat(R.F2, GEO::L_F2);
    if constexpr (TWR) lds_ct16(R.tw, GEO::L_TW);
    const i32 lane = B::lane();
    const i32 c = lane & 31, hi = lane >> 5;
    const i32 i16 = lane & 15, g16 = (lane >> 4) & 1;
    const i32 n3b = (g16 * 16) % GEO::N3;
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
      for (int rho = 0; rho < 2; rho++) {
        i32 U = hi * 4 + (16 * ms + 8 * rho) + (i16 >> 2);
        i32 sU = U / GEO::N2, n2 = U % GEO::N2;
        i32 row = sU * GEO::SV + ((g16 * 16) / GEO::N3);
        R.roff[ms][rho] = e_off<GEO, i32>(row, n2 * GEO::N3 + n3b + (i16 & 3) * 4) + un.eb;
      }
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 V = hi * 4 + 8 * rq;
      i32 row = (c / GEO::N2) * GEO::SV + V / GEO::N3;
      R.woff[rq] = e_off<GEO, i32>(row, (c % GEO::N2) * GEO::N3 + V % GEO::N3) + un.eb;
    }
  }

  // Inner-only multi-pass form (fft 2048 = 2 passes of the 32 x 32 kernel, struct Pass): with m = 32 n2 + n3 the pass
  // factor W_N^{m k0} = W_{N/32}^{n2 k0} W_N^{n3 k0}: its n2 part rides in the stage-a DFT matrix (and, conjugated, in
  // the last inverse matrix), its n3 part in the two inner twiddle tables.  Per pass: [Fa | Finv | tw | tw2] in LDS behind
  // the single-pass tables (HostPlan tabs.ipass, copied by setup_tables_ipass).
  static constexpr int IPASS_BYTES = 2 * 6144 + 2 * 8192;
  static constexpr int L_IPASS = GEO::LDS_BYTES;
  struct InnerPass { Mat Fa, Finv; CT16 tw; int tw2_off; int base; };
  static FFC_FN void load_inner_pass(InnerPass& ip, int k0) {
    const int base = L_IPASS + k0 * IPASS_BYTES;
    lds_mat(ip.Fa, base);
    lds_mat(ip.Finv, base + 6144);
    lds_ct16(ip.tw, base + 12288);
    ip.tw2_off = base + 12288 + 8192;
    ip.base = base;
  }
  // lean form (IPL, round 6): only the LDS offsets; the pass's two matrices and its twiddle table are read from LDS where they are used
  // (tile_fwd / tile_inv <.., IPL>).  The backward of fft 2048 kept all 80 registers of a pass next to its dk_f sums and spilled 34 - 43 of them.
  static FFC_FN void load_inner_pass_lean(InnerPass& ip, int k0) {
    ip.base = L_IPASS + k0 * IPASS_BYTES;
    ip.tw2_off = ip.base + 12288 + 8192;
  }
  static FFC_FN void setup_tables_ipass(const uint8_t* tab, const PlanTabs& t, int R) {
    for (int k0 = 0; k0 < R; k0++) copy_tab(tab + t.ipass[k0], L_IPASS + k0 * IPASS_BYTES, IPASS_BYTES);
    B::barrier();
  }

  static FFC_FN void load_tile_op(int tau, Op& op, Unit un, const InnerRegs& R, int doff = 0) {
    const int trow = tau * (GEO::G * GEO::Mi * 2) + doff;     // doff: E of the partner unit (inner_tile2x)
    if (B::HAS_TR) {
#pragma unroll
      for (int ms = 0; ms < 2; ms++)
#pragma unroll
        for (int rho = 0; rho < 2; rho++) {
          i32 off = R.roff[ms][rho] + trow;
          U2 vr = B::lds_r64_tr(off), vi = B::lds_r64_tr(off + GEO::PLANE);
          op.r[ms][2 * rho] = vr.x; op.r[ms][2 * rho + 1] = vr.y;
          op.i[ms][2 * rho] = vi.x; op.i[ms][2 * rho + 1] = vi.y;
        }
#pragma unroll
      for (int ms = 0; ms < 2; ms++) { B::pin(op.r[ms]); B::pin(op.i[ms]); }
    } else {
      const i32 lane = B::opaque(B::lane());
      const i32 c = lane & 31, hi = lane >> 5;
      const i32 sV = c / GEO::N3, n3 = c % GEO::N3;
#pragma unroll
      for (int ms = 0; ms < 2; ms++)
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 wr[2], wi[2];
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {
            int e = 2 * d + hf;
            i32 U = hi * 4 + (16 * ms + 8 * (e >> 2) + (e & 3));
            i32 sU = U / GEO::N2, n2 = U % GEO::N2;
            i32 row = sU * GEO::SV + sV + tau * GEO::G;
            i32 off = e_off<GEO, i32>(row, n2 * GEO::N3 + n3) + un.eb + doff;
            wr[hf] = B::lds_r16(off);
            wi[hf] = B::lds_r16(off + GEO::PLANE);
          }
          op.r[ms][d] = wr[0] | (wr[1] << 16);
          op.i[ms][d] = wi[0] | (wi[1] << 16);
        }
    }
  }

  // forward half: E tile -> Z = s_fwd*FFT in layout [V'=(sV,k3) regs][U'=(sU,k2) lanes]
  // TWR: inner twiddle table resident in registers (R.tw); false -> re-read from LDS at each use (saves 32
  // VGPRs in the register-heavy backward kernels)
  // FFC_FOLD_TW (round 5): the outer twiddle W_N^{m k1}, m = 32 n2 + n3, is the product of a factor on n2 and one on n3,
  // and each inner stage contracts (forward) resp. produces (inverse) exactly one of the two indices -- so both factors fold into
  // the stages' DFT matrices, one 6 KB operand table per (stage, tile k1) from L2 (PlanTabs::fold, 768 KB per plan), and NO elementwise
  // outer twiddle is left: no chain (v_sin / v_cos), no 16 complex multiplies per tile and direction.  s_fwd / s_inv ride in the
  // stage-a matrices.  Geo<32,32,32>, single pass.  `fold` = plan blob + PlanTabs::fold.
  // FFC_FOLD_TW (ffc_plan.h): 1 = the forward kernels of fft 16384 (default), 2 = also fft 32768 (forward + saved-spectra backward)
  static constexpr bool CAN_FOLD = ((FFC_FOLD_TW >= 1 && GEO::N1 == 16) || (FFC_FOLD_TW >= 2 && GEO::N1 == 32)) && GEO::N2 == 32 && GEO::N3 == 32;
  template <bool TWR = true, bool IP = false, bool FOLD = false, bool IPL = false>
  static FFC_FN void tile_fwd(int tau, const InnerRegs& R, Unit un, A16& re, A16& im, const InnerPass* ip = nullptr, const uint8_t* fold = nullptr,
                              const Mat2* fa_pre = nullptr) {
    if constexpr (FOLD) {
      const i32 lane = B::opaque(B::lane());
      Mat2 FA, FB;
      if (fa_pre) FA = *fa_pre;                 // requested by the caller during the previous tile (one stage of lookahead)
      else load_mat2_issue(FA, fold + (0 * GEO::NT + tau) * 6144, lane);
      load_mat2_issue(FB, fold + (1 * GEO::NT + tau) * 6144, lane);
      Op op;
      load_tile_op(tau, op, un, R);
      re = B::a16_zero(); im = B::a16_zero();
      cmm2<true>(re, im, op, FA);
      if constexpr (TWR) cmul(re, im, R.tw);
      else cmul_lds<false>(re, im, GEO::L_TW);
      to_op(re, im, op);
      re = B::a16_zero(); im = B::a16_zero();
      cmm2<false>(re, im, op, FB);
      return;
    }
    Op op;
    load_tile_op(tau, op, un, R);
    // stage a: contract n2 (A-form) -> [V=(sV,n3) regs][U'=(sU,k2) lanes]
    re = B::a16_zero(); im = B::a16_zero();
    if constexpr (IP && IPL) {
      Mat Fa;
      lds_mat(Fa, ip->base);
      cmm<false, true>(re, im, op, Fa);
      cmul_lds<false>(re, im, ip->base + 12288);
    } else if constexpr (IP) {
      cmm<false, true>(re, im, op, ip->Fa);
      cmul(re, im, ip->tw);
    } else {
      cmm<false, true>(re, im, op, R.F2);
      if constexpr (TWR) {
        cmul(re, im, R.tw);
      } else {
        cmul_lds<false>(re, im, GEO::L_TW);
      }
    }
    to_op(re, im, op);
    // stage b: contract n3 (B-form) -> [V'=(sV,k3) regs][U' lanes]
    re = B::a16_zero(); im = B::a16_zero();
    if constexpr (GEO::N3 != GEO::N2) {
      Mat F3;
      lds_mat(F3, GEO::L_F3);
      cmm<false, false>(re, im, op, F3);
    } else {
      cmm<false, false>(re, im, op, R.F2);
    }
  }
  // inverse half: spectrum tile (same layout) -> E tile, incl. the outer inverse twiddle
  template <bool TWR = true, bool RP = false, bool IP = false, bool FOLD = false, bool IPL = false>
  static FFC_FN void tile_inv(float s_inv, int tau, const InnerRegs& R, Unit un, A16& re, A16& im, int dbg = 0, Pass ps = Pass(),
                              const InnerPass* ip = nullptr, const uint8_t* fold = nullptr, const Mat2* g_pre = nullptr) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
    if constexpr (FOLD) {
      static_assert(!RP && !IP && CAN_FOLD, "folded outer twiddle: single-pass fft 16384 / 32768");
      Mat2 GB, GA;                  // conjugation and the outer inverse factors (+ s_inv) are in the tables: plain products
      if (g_pre) { GB = g_pre[0]; GA = g_pre[1]; }      // requested by the caller behind the forward half
      else {
        load_mat2_issue(GB, fold + (2 * GEO::NT + tau) * 6144, lane);
        load_mat2_issue(GA, fold + (3 * GEO::NT + tau) * 6144, lane);
      }
      Op op;
      to_op(re, im, op);
      re = B::a16_zero(); im = B::a16_zero();
      cmm2<true>(re, im, op, GB);
      if constexpr (TWR) cmul_conj(re, im, R.tw);
      else cmul_lds<true>(re, im, GEO::L_TW);
      to_op(re, im, op);
      re = B::a16_zero(); im = B::a16_zero();
      cmm2<true>(re, im, op, GA);
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        i32 off = R.woff[rq] + tau * (GEO::G * GEO::Mi * 2);
        U2 vr, vi;
        vr.x = B::template pack<DT>(re[4 * rq], re[4 * rq + 1]);
        vr.y = B::template pack<DT>(re[4 * rq + 2], re[4 * rq + 3]);
        vi.x = B::template pack<DT>(im[4 * rq], im[4 * rq + 1]);
        vi.y = B::template pack<DT>(im[4 * rq + 2], im[4 * rq + 3]);
        B::lds_w64(off, vr);
        B::lds_w64(off + GEO::PLANE, vi);
      }
      return;
    }
    Op op;
    to_op(re, im, op);
    // inverse stage b: contract k3 (A-form, conj) -> [U' regs][V''=(sV,n3) lanes]
    re = B::a16_zero(); im = B::a16_zero();
    if constexpr (GEO::N3 != GEO::N2) {
      Mat F3;
      lds_mat(F3, GEO::L_F3);
      cmm<true, true>(re, im, op, F3);
    } else {
      cmm<true, true>(re, im, op, R.F2);
    }
    if constexpr (IP) {
      cmul_lds<false>(re, im, ip->tw2_off);
    } else if constexpr (GEO::TW2_SEP) {
      cmul_lds<false>(re, im, GEO::L_TW2);
    } else if constexpr (TWR) {
      cmul_conj(re, im, R.tw);
    } else {
      cmul_lds<true>(re, im, GEO::L_TW);
    }
    to_op(re, im, op);
    // inverse stage a: contract k2 (A-form, conj) -> [V'' regs][U''=(sU,n2) lanes]
    re = B::a16_zero(); im = B::a16_zero();
    if constexpr (IP && IPL) {
      Mat Fi;
      lds_mat(Fi, ip->base + 6144);
      cmm<true, true>(re, im, op, Fi);
    } else if constexpr (IP) cmm<true, true>(re, im, op, ip->Finv);
    else cmm<true, true>(re, im, op, R.F2);
    // outer inverse twiddle s_inv * W_N^{-(n2*N3+n3)*k1}, generated on the fly (v_sin/v_cos take
    // revolutions; the integer phase m*k1 mod N is exact), so no table traffic in the tile loop
    if constexpr (GEO::OUTER) {
      // registers <-> V = (sV, n3) = 4*hi + {0..3} + 8*{0..3}; lane <-> (sU, n2); E row k1 = tau*G + sU*SV + sV
      const i32 sUl = c / GEO::N2, mlane = (c % GEO::N2) * GEO::N3;
      if (dbg & 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          i32 V = hi * 4 + ((r & 3) + 8 * (r >> 2));
          i32 k1 = sUl * GEO::SV + V / GEO::N3 + tau * GEO::G;
          f32 tr, ti;
          cis_rev(B::mul24(mlane + V % GEO::N3, k1), 1.0f, &tr, &ti);
          tr = tr * s_inv; ti = ti * s_inv;
          f32 xr = re[r], xi = im[r];
          re[r] = xr * tr - xi * ti;
          im[r] = xr * ti + xi * tr;
        }
      } else if constexpr (GEO::N3 == 32) {
        // both register halves lie in the same E row k1 (n3 and n3 + 16): one chain (twiddle16)
        i32 k1 = sUl * GEO::SV + tau * GEO::G;
        i32 n30 = hi * 4;
        if constexpr (RP) {
          i32 kk = k1 * ps.R + ps.k0;
          twiddle16(re, im, B::mul24(mlane + n30, kk), kk, 1.0f, s_inv, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
        } else {
          twiddle16(re, im, B::mul24(mlane + n30, k1), k1, 1.0f, s_inv);
        }
      } else
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int sV = (16 * half) / GEO::N3;
        i32 k1 = sUl * GEO::SV + (sV + tau * GEO::G);
        i32 n30 = hi * 4 + ((16 * half) % GEO::N3);
        F2 tr[4], ti[4];
        // conj(W^{m k1}) = cos + i sin; multi-pass: k1 -> k0 + R k1 on the N-point circle
        if constexpr (RP) {
          i32 kk = k1 * ps.R + ps.k0;
          chain8p(B::mul24(mlane + n30, kk), kk, 1.0f, s_inv, tr, ti, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
        } else {
          chain8p(B::mul24(mlane + n30, k1), k1, 1.0f, s_inv, tr, ti);
        }
        apply8(re, im, half, tr, ti);
      }
    }
    // write back in place: lane <-> (sU,n2), regs <-> (sV,n3); r&3 = 4 consecutive n3
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 off = R.woff[rq] + tau * (GEO::G * GEO::Mi * 2);
      U2 vr, vi;
      vr.x = B::template pack<DT>(re[4 * rq], re[4 * rq + 1]);
      vr.y = B::template pack<DT>(re[4 * rq + 2], re[4 * rq + 3]);
      vi.x = B::template pack<DT>(im[4 * rq], im[4 * rq + 1]);
      vi.y = B::template pack<DT>(im[4 * rq + 2], im[4 * rq + 3]);
      B::lds_w64(off, vr);
      B::lds_w64(off + GEO::PLANE, vi);
    }
  }

  struct KfRegs { U4 v[4]; };
  // A spectrum tile in global memory: (re, im)-interleaved dtype pairs in the k_f tile layout (16-byte accesses, 1 KiB
  // per wave instruction).  Used for the backward kernels' scratch and for the spectra the forward pass saves.
  static FFC_FN void z_store(void* zs, int tau, const A16& re, const A16& im, bool nt = false) {
#if defined(FFC_KO) && (FFC_KO & 64)
    return;        // knock-out timing experiment: no spectrum scratch traffic (results wrong)
#endif
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      U4 v;
      v.x = B::template pack<DT>(re[4 * rq], im[4 * rq]);         v.y = B::template pack<DT>(re[4 * rq + 1], im[4 * rq + 1]);
      v.z = B::template pack<DT>(re[4 * rq + 2], im[4 * rq + 2]); v.w = B::template pack<DT>(re[4 * rq + 3], im[4 * rq + 3]);
      if (nt) B::g_w128_nt(zs, ((hi + (tau * 8 + 2 * rq)) * 32 + c), v, B::ptrue());
      else B::g_w128(zs, ((hi + (tau * 8 + 2 * rq)) * 32 + c), v, B::ptrue());
    }
  }
  static FFC_FN void z_load(const void* zs, int tau, KfRegs& z, bool nt = false) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#if defined(FFC_KO) && (FFC_KO & 64)
    for (int rq = 0; rq < 4; rq++) { z.v[rq].x = B::as_u32(B::i2f(c + tau)); z.v[rq].y = z.v[rq].x; z.v[rq].z = z.v[rq].x; z.v[rq].w = z.v[rq].x; }
    return;
#endif
    if (nt) {
#pragma unroll
      for (int rq = 0; rq < 4; rq++) z.v[rq] = B::g_r128_nt(zs, ((hi + (tau * 8 + 2 * rq)) * 32 + c));
      return;
    }
#pragma unroll
    for (int rq = 0; rq < 4; rq++) z.v[rq] = B::g_r128(zs, ((hi + (tau * 8 + 2 * rq)) * 32 + c));
  }
  // saved spectrum of pair p of head h (ConvArgs::zsave layout)
  static FFC_FN uint8_t* z_slot(void* base, int h, int npair, int p) {
    return (uint8_t*)base + ((int64_t)h * npair + p) * ((int64_t)GEO::N * 4);
  }
  // single-tile sizes (fft <= 2048): [H][tiles of G pairs][R passes] slots of 1024 complex values
  static FFC_FN uint8_t* z_slot_small(void* base, int h, int npair, int q, int R, int k0) {
    const int ntile = (npair + GEO::G - 1) / GEO::G;
    return (uint8_t*)base + (((int64_t)h * ntile + q) * R + k0) * 4096;
  }
  // multi-pass sizes: [H][npair][R passes][M]
  static FFC_FN uint8_t* z_slot_rp(void* base, int h, int npair, int p, int R, int k0) {
    return (uint8_t*)base + (((int64_t)h * npair + p) * R + k0) * ((int64_t)GEO::N * 4);
  }
  static FFC_FN void load_kf(const ConvArgs& a, int h, int tau, KfRegs& k) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
    const uint8_t* kfh = (const uint8_t*)a.kf + (int64_t)h * (GEO::NT * 1024 * 4);
#if defined(FFC_KO) && (FFC_KO & 4)
    for (int rq = 0; rq < 4; rq++) { k.v[rq].x = B::as_u32(B::i2f(c + tau)); k.v[rq].y = k.v[rq].x; k.v[rq].z = k.v[rq].x; k.v[rq].w = k.v[rq].x; }
    return;
#endif
    if (a.flags & 2) {      // tuning flag: k_f as a streaming (non-temporal) read
#pragma unroll
      for (int rq = 0; rq < 4; rq++) k.v[rq] = B::g_r128_nt(kfh, ((hi + (tau * 8 + 2 * rq)) * 32 + c));
      return;
    }
#pragma unroll
    for (int rq = 0; rq < 4; rq++) k.v[rq] = B::g_r128(kfh, ((hi + (tau * 8 + 2 * rq)) * 32 + c));
  }
  // SZ: the tile's spectrum is kept for the backward pass at zs (single-tile sizes: one 4 KB slot per tile and pass, z_slot_small)
  template <bool IP = false, bool SZ = false>
  static FFC_FN void inner_tile(const ConvArgs& a, int tau, const InnerRegs& R, Unit un, const KfRegs& kf, const InnerPass* ip = nullptr,
                                uint8_t* zs = nullptr) {
    A16 re, im;
    tile_fwd<true, IP>(tau, R, un, re, im, ip);
    if constexpr (SZ) {      // (single-tile sizes: zs may be null -- the gated forward that keeps only the output before the postgate, ConvArgs::yraw)
      if (GEO::OUTER || zs) z_store(zs, 0, re, im, FFC_Z_STREAM);
    }
    // (x) k_f
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      u32 wv[4] = {kf.v[rq].x, kf.v[rq].y, kf.v[rq].z, kf.v[rq].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32 kr = B::template unpack_lo<DT>(wv[q]), ki = B::template unpack_hi<DT>(wv[q]);
        if (a.conj_kf) ki = B::fconst(0.f) - ki;
        int r = 4 * rq + q;
        f32 x = re[r], y = im[r];
        re[r] = x * kr - y * ki;
        im[r] = x * ki + y * kr;
      }
    }
    tile_inv<true, false, IP>(a.s_inv, tau, R, un, re, im, 0, Pass(), ip);
  }

  // Two tiles processed in lock-step: their chains are independent, so the MFMAs of one tile execute while
  // the twiddle / conversion VALU work of the other one issues (a single tile alternates MFMA-only and
  // VALU-only stretches and exposes the MFMA dependency latency each time).
  // NOCONJ (compile time): the caller never asks for conj(k_f) (the spectrum-saving training forward): no sign multiply
  template <bool NOCONJ = false>
  static FFC_FN void kf_mul(const ConvArgs& a, const KfRegs& kf, A16& re, A16& im) {
    CT16 k;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      u32 wv[4] = {kf.v[rq].x, kf.v[rq].y, kf.v[rq].z, kf.v[rq].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        k.re[4 * rq + q] = B::template unpack_lo<DT>(wv[q]);
        k.im[4 * rq + q] = B::template unpack_hi<DT>(wv[q]);
      }
    }
    k.im = B::a16_scale(k.im, a.conj_kf ? -1.0f : 1.0f);
    cmul(re, im, k);
  }
  // ---- frequency-sparse phase B (SP): only the spectrum rows k3 = 0..3 (accumulator registers 0..3 of lane half 0) and
  // 28..31 (registers 12..15 of lane half 1) can be non-zero, because k_f is zero everywhere else.  Every lane works on its
  // registers {0..3, 12..15} (the unused half of the set multiplies zeros of k_f), the other eight are never touched again.
  struct KfRegsSp { U4 v[2]; };      // rows rq = 0 and rq = 3 of KfRegs
  static FFC_FN void load_kf_sp(const ConvArgs& a, int h, int tau, KfRegsSp& k) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
    const uint8_t* kfh = (const uint8_t*)a.kf + (int64_t)h * (GEO::NT * 1024 * 4);
    k.v[0] = B::g_r128(kfh, ((hi + (tau * 8 + 0)) * 32 + c));
    k.v[1] = B::g_r128(kfh, ((hi + (tau * 8 + 6)) * 32 + c));
  }
  static FFC_FN void kf_mul_sp(const ConvArgs& a, const KfRegsSp& kf, A16& re, A16& im) {
    const float sg = a.conj_kf ? -1.0f : 1.0f;
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      u32 wv[4] = {kf.v[s2].x, kf.v[s2].y, kf.v[s2].z, kf.v[s2].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 12 * s2 + q;
        f32 kr = B::template unpack_lo<DT>(wv[q]), ki = B::template unpack_hi<DT>(wv[q]) * sg;
        f32 x = re[r], y = im[r];
        re[r] = x * kr - y * ki;
        im[r] = x * ki + y * kr;
      }
    }
  }
  // the one K-step of the sparse inverse stage b: slots e < 4 = this lane half's four live rows, slots e >= 4 = zero
  static FFC_FN void to_op_sp(const A16& re, const A16& im, Op& o) {
    const pred up = (B::opaque(B::lane()) >> 5) >= 1;
#pragma unroll
    for (int d = 0; d < 2; d++) {
      o.r[0][d] = B::sel(up, B::template pack<DT>(re[12 + 2 * d], re[13 + 2 * d]), B::template pack<DT>(re[2 * d], re[2 * d + 1]));
      o.i[0][d] = B::sel(up, B::template pack<DT>(im[12 + 2 * d], im[13 + 2 * d]), B::template pack<DT>(im[2 * d], im[2 * d + 1]));
      o.r[0][2 + d] = B::uconst(0); o.i[0][2 + d] = B::uconst(0);
      o.r[1][d] = B::uconst(0); o.i[1][d] = B::uconst(0); o.r[1][2 + d] = B::uconst(0); o.i[1][2 + d] = B::uconst(0);
    }
  }
  static FFC_FN void lds_mat_sp(Mat& m) {
    const i32 lane = B::lane();
#pragma unroll
    for (int q = 0; q < 3; q++) {
      U4 v = B::lds_r128(lane * 16 + (GEO::L_FS + q * 1024));
      m.w[0][q] = B::w4(v.x, v.y, v.z, v.w);
      m.w[1][q] = B::w4(B::uconst(0), B::uconst(0), B::uconst(0), B::uconst(0));
    }
#pragma unroll
    for (int q = 0; q < 3; q++) B::pin(m.w[0][q]);
  }
  // two tiles of the pair in lock-step, sparse spectrum (the dense form is inner_tile2)
  static FFC_FN void inner_tile2_sp(const ConvArgs& a, int h, int tauA, const InnerRegs& R, const Mat& Fs, Unit un) {
    static_assert(GEO::HAS_SP, "sparse phase B: 32-point inner digits");
    const int tauB = tauA + 1;
    KfRegsSp kfA, kfB;
    load_kf_sp(a, h, tauA, kfA);
    load_kf_sp(a, h, tauB, kfB);
    Op opA, opB;
    load_tile_op(tauA, opA, un, R);
    load_tile_op(tauB, opB, un, R);
    A16 reA, imA, reB, imB;
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<false, true>(reA, imA, opA, R.F2);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<false, true>(reB, imB, opB, R.F2);
    cmul(reA, imA, R.tw); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<false, false>(reA, imA, opA, R.F2);
    cmul(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<false, false>(reB, imB, opB, R.F2);
    // (x) k_f on the live rows, inverse stage b with ONE K-step
    kf_mul_sp(a, kfA, reA, imA); to_op_sp(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<true, true>(reA, imA, opA, Fs, 1);
    kf_mul_sp(a, kfB, reB, imB); to_op_sp(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<true, true>(reB, imB, opB, Fs, 1);
    cmul_conj(reA, imA, R.tw); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<true, true>(reA, imA, opA, R.F2);
    cmul_conj(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<true, true>(reB, imB, opB, R.F2);
    oi_twiddle<false>(a.s_inv, tauA, reA, imA);
    tile_store(tauA, R, reA, imA);
    oi_twiddle<false>(a.s_inv, tauB, reB, imB);
    tile_store(tauB, R, reB, imB);
  }
  template <bool RP = false>
  static FFC_FN void oi_twiddle(float s_inv, int tau, A16& re, A16& im, Pass ps = Pass()) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
    const i32 sUl = c / GEO::N2, mlane = (c % GEO::N2) * GEO::N3;
    if constexpr (GEO::N3 == 32) {       // one chain for both register halves (twiddle16)
      i32 k1 = sUl * GEO::SV + tau * GEO::G;
      i32 n30 = hi * 4;
      if constexpr (RP) {
        i32 kk = k1 * ps.R + ps.k0;
        twiddle16(re, im, B::mul24(mlane + n30, kk), kk, 1.0f, s_inv, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
      } else {
        twiddle16(re, im, B::mul24(mlane + n30, k1), k1, 1.0f, s_inv);
      }
      return;
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int sV = (16 * half) / GEO::N3;
      i32 k1 = sUl * GEO::SV + (sV + tau * GEO::G);
      i32 n30 = hi * 4 + ((16 * half) % GEO::N3);
      F2 tr[4], ti[4];
      if constexpr (RP) {
        i32 kk = k1 * ps.R + ps.k0;
        chain8p(B::mul24(mlane + n30, kk), kk, 1.0f, s_inv, tr, ti, GEO::N * ps.R - 1, 1.0f / (float)(GEO::N * ps.R));
      } else {
        chain8p(B::mul24(mlane + n30, k1), k1, 1.0f, s_inv, tr, ti);
      }
      apply8(re, im, half, tr, ti);
    }
  }
  static FFC_FN void tile_store(int tau, const InnerRegs& R, const A16& re, const A16& im, int doff = 0) {
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 off = R.woff[rq] + tau * (GEO::G * GEO::Mi * 2) + doff;
      U2 vr, vi;
      vr.x = B::template pack<DT>(re[4 * rq], re[4 * rq + 1]);
      vr.y = B::template pack<DT>(re[4 * rq + 2], re[4 * rq + 3]);
      vi.x = B::template pack<DT>(im[4 * rq], im[4 * rq + 1]);
      vi.y = B::template pack<DT>(im[4 * rq + 2], im[4 * rq + 3]);
      B::lds_w64(off, vr);
      B::lds_w64(off + GEO::PLANE, vi);
    }
  }
  // SZ (compile time: phase B must stay one basic block): store both tiles' spectra at zs (ConvArgs::zsave)
  // FFC_FOLD_TW form of inner_tile2 (see tile_fwd): every stage of either tile multiplies by its own per-(stage, tile) matrix from L2 with the
  // outer twiddle folded in; the loads of a stage's two matrices are requested one stage ahead.  No oi_twiddle, and phase A ran without
  // its twiddle (outer_stage<.., NOTW>).  LDSTW: inner twiddle streamed from LDS instead of the 32 resident registers.
#ifndef FFC_FOLD_LDSTW
#define FFC_FOLD_LDSTW 0
#endif
#ifndef FFC_FOLD_FWD
#define FFC_FOLD_FWD 1      // (with FFC_FOLD_TW) 0: only the saved-spectra backward folds, the forward kernels keep the chains
#endif
  template <bool SZ>
  static FFC_FN void inner_tile2_fold(const ConvArgs& a, int h, int tauA, const InnerRegs& R, Unit un, uint8_t* zs) {
    static_assert(CAN_FOLD, "folded outer twiddle: fft 32768 geometry");
    const int tauB = tauA + 1;
    const i32 lane = B::opaque(B::lane());
    const uint8_t* fold = a.tab + a.t.fold;
    Mat2 FAa, FAb, FBa, FBb;
    load_mat2_issue(FAa, fold + (0 * GEO::NT + tauA) * 6144, lane);
    load_mat2_issue(FAb, fold + (0 * GEO::NT + tauB) * 6144, lane);
    KfRegs kfA, kfB;
    load_kf(a, h, tauA, kfA);
    load_kf(a, h, tauB, kfB);
    Op opA, opB;
    load_tile_op(tauA, opA, un, R);
    load_tile_op(tauB, opB, un, R);
    load_mat2_issue(FBa, fold + (1 * GEO::NT + tauA) * 6144, lane);
    load_mat2_issue(FBb, fold + (1 * GEO::NT + tauB) * 6144, lane);
    A16 reA, imA, reB, imB;
    auto tw_fwd = [&](A16& re, A16& im) { if constexpr (FFC_FOLD_LDSTW != 0) cmul_lds<false>(re, im, GEO::L_TW); else cmul(re, im, R.tw); };
    auto tw_inv = [&](A16& re, A16& im) { if constexpr (FFC_FOLD_LDSTW != 0) cmul_lds<true>(re, im, GEO::L_TW); else cmul_conj(re, im, R.tw); };
    // stage a
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm2<true>(reA, imA, opA, FAa);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm2<true>(reB, imB, opB, FAb);
    tw_fwd(reA, imA); to_op(reA, imA, opA);
    Mat2 GBa, GBb;
    load_mat2_issue(GBa, fold + (2 * GEO::NT + tauA) * 6144, lane);
    load_mat2_issue(GBb, fold + (2 * GEO::NT + tauB) * 6144, lane);
    // stage b
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm2<false>(reA, imA, opA, FBa);
    tw_fwd(reB, imB); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm2<false>(reB, imB, opB, FBb);
    if constexpr (SZ) { z_store(zs, tauA, reA, imA, FFC_Z_STREAM); z_store(zs, tauB, reB, imB, FFC_Z_STREAM); }
    // (x) k_f, inverse stage b
    kf_mul<SZ>(a, kfA, reA, imA); to_op(reA, imA, opA);
    Mat2 GAa, GAb;
    load_mat2_issue(GAa, fold + (3 * GEO::NT + tauA) * 6144, lane);
    load_mat2_issue(GAb, fold + (3 * GEO::NT + tauB) * 6144, lane);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm2<true>(reA, imA, opA, GBa);
    kf_mul<SZ>(a, kfB, reB, imB); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm2<true>(reB, imB, opB, GBb);
    // inverse inner twiddle, inverse stage a (outer inverse twiddle and s_inv are in GA)
    tw_inv(reA, imA); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm2<true>(reA, imA, opA, GAa);
    tw_inv(reB, imB); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm2<true>(reB, imB, opB, GAb);
    tile_store(tauA, R, reA, imA);
    tile_store(tauB, R, reB, imB);
  }
  // one tile at a time (FFC_FOLD_ONE: the L <= N/2 forward kernels, whose next pair's rows wait in 32 registers across phase B -- with two
  // tiles' matrices in flight they spilled 60 .. 170 registers)
#ifndef FFC_FOLD_ONE
#define FFC_FOLD_ONE 2      // 0: two tiles in lock-step everywhere, 1: one tile at a time everywhere, 2: one at a time in the L <= N/2 kernels only
#endif
  template <bool SZ>
  static FFC_FN void inner_tile1_fold(const ConvArgs& a, int h, int tau, const InnerRegs& R, Unit un, uint8_t* zs) {
    static_assert(CAN_FOLD, "folded outer twiddle: fft 32768 geometry");
    const i32 lane = B::opaque(B::lane());
    const uint8_t* fold = a.tab + a.t.fold;
    Mat2 FA, FB, GB, GA;
    load_mat2_issue(FA, fold + (0 * GEO::NT + tau) * 6144, lane);
    KfRegs kf;
    load_kf(a, h, tau, kf);
    Op op;
    load_tile_op(tau, op, un, R);
    load_mat2_issue(FB, fold + (1 * GEO::NT + tau) * 6144, lane);
    A16 re, im;
    re = B::a16_zero(); im = B::a16_zero();
    cmm2<true>(re, im, op, FA);
    load_mat2_issue(GB, fold + (2 * GEO::NT + tau) * 6144, lane);
    if constexpr (FFC_FOLD_LDSTW != 0) cmul_lds<false>(re, im, GEO::L_TW); else cmul(re, im, R.tw);
    to_op(re, im, op);
    re = B::a16_zero(); im = B::a16_zero();
    cmm2<false>(re, im, op, FB);
    load_mat2_issue(GA, fold + (3 * GEO::NT + tau) * 6144, lane);
    if constexpr (SZ) z_store(zs, tau, re, im, FFC_Z_STREAM);
    kf_mul<SZ>(a, kf, re, im); to_op(re, im, op);
    re = B::a16_zero(); im = B::a16_zero();
    cmm2<true>(re, im, op, GB);
    if constexpr (FFC_FOLD_LDSTW != 0) cmul_lds<true>(re, im, GEO::L_TW); else cmul_conj(re, im, R.tw);
    to_op(re, im, op);
    re = B::a16_zero(); im = B::a16_zero();
    cmm2<true>(re, im, op, GA);
    tile_store(tau, R, re, im);
  }
  template <bool RP = false, bool SZ = false, bool FOLD = false, bool HALFK = false>
  static FFC_FN void inner_tile2(const ConvArgs& a, int h, int tauA, const InnerRegs& R, Unit un, Pass ps = Pass(), uint8_t* zs = nullptr) {
    static_assert(GEO::N3 == GEO::N2 && GEO::OUTER, "inner_tile2: fused sizes with one inner matrix");
    if constexpr (FOLD) {
      if constexpr (FFC_FOLD_ONE == 1 || (FFC_FOLD_ONE == 2 && HALFK)) { inner_tile1_fold<SZ>(a, h, tauA, R, un, zs); inner_tile1_fold<SZ>(a, h, tauA + 1, R, un, zs); }
      else inner_tile2_fold<SZ>(a, h, tauA, R, un, zs);
      return;
    }
    const int tauB = tauA + 1;
    KfRegs kfA, kfB;
    load_kf(a, h, tauA, kfA);
    load_kf(a, h, tauB, kfB);
    Op opA, opB;
    load_tile_op(tauA, opA, un, R);
    load_tile_op(tauB, opB, un, R);
    A16 reA, imA, reB, imB;
    // stage a
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<false, true>(reA, imA, opA, R.F2);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<false, true>(reB, imB, opB, R.F2);
    cmul(reA, imA, R.tw); to_op(reA, imA, opA);
    // stage b (A) while B's twiddle issues
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<false, false>(reA, imA, opA, R.F2);
    cmul(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<false, false>(reB, imB, opB, R.F2);
    if constexpr (SZ) { z_store(zs, tauA, reA, imA, FFC_Z_STREAM); z_store(zs, tauB, reB, imB, FFC_Z_STREAM); }     // spectrum kept for the backward pass
    // (x) k_f, inverse stage b
    kf_mul<SZ>(a, kfA, reA, imA); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<true, true>(reA, imA, opA, R.F2);
    kf_mul<SZ>(a, kfB, reB, imB); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<true, true>(reB, imB, opB, R.F2);
    // inverse twiddle, inverse stage a
    cmul_conj(reA, imA, R.tw); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<true, true>(reA, imA, opA, R.F2);
    cmul_conj(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<true, true>(reB, imB, opB, R.F2);
    // outer inverse twiddle + write back
    oi_twiddle<RP>(a.s_inv, tauA, reA, imA, ps);
    tile_store(tauA, R, reA, imA);
    oi_twiddle<RP>(a.s_inv, tauB, reB, imB, ps);
    tile_store(tauB, R, reB, imB);
  }

  // Cross-unit form (fft 8192 / 16384: two or four units per workgroup): tile tau of TWO units (two pairs of the same head)
  // in lock-step instead of two tiles of one unit.  Both tiles meet the same k_f tile and the same outer inverse twiddle
  // W_N^{-m k1}: one k_f load + unpack and one twiddle chain (4 v_sin/v_cos pairs + the chain multiplies) serve both.
  // R holds the offsets of the first unit of the group, the partner's E lies EBYTES behind it.
  // FFC_FOLD_TW form of inner_tile2x: the two units' tiles share the tile's four folded matrices (half the matrix bytes per pair of the
  // one-unit sizes), each requested one stage ahead
  template <bool SZ>
  static FFC_FN void inner_tile2x_fold(const ConvArgs& a, int h, int tau, const InnerRegs& R, Unit un, uint8_t* zsA, uint8_t* zsB) {
    static_assert(CAN_FOLD && GEO::UPW >= 2, "folded outer twiddle, two units per workgroup");
    constexpr int DB = GEO::EBYTES;
    const i32 lane = B::opaque(B::lane());
    const uint8_t* fold = a.tab + a.t.fold;
    Mat2 FA, FB, GB, GA;
    load_mat2_issue(FA, fold + (0 * GEO::NT + tau) * 6144, lane);
    KfRegs kf;
    load_kf(a, h, tau, kf);
    Op opA, opB;
    load_tile_op(tau, opA, un, R);
    load_tile_op(tau, opB, un, R, DB);
    load_mat2_issue(FB, fold + (1 * GEO::NT + tau) * 6144, lane);
    A16 reA, imA, reB, imB;
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm2<true>(reA, imA, opA, FA);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm2<true>(reB, imB, opB, FA);
    load_mat2_issue(GB, fold + (2 * GEO::NT + tau) * 6144, lane);
    cmul(reA, imA, R.tw); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm2<false>(reA, imA, opA, FB);
    cmul(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm2<false>(reB, imB, opB, FB);
    load_mat2_issue(GA, fold + (3 * GEO::NT + tau) * 6144, lane);
    if constexpr (SZ) { z_store(zsA, tau, reA, imA, FFC_Z_STREAM); z_store(zsB, tau, reB, imB, FFC_Z_STREAM); }
    {
      CT16 k;
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        u32 wv[4] = {kf.v[rq].x, kf.v[rq].y, kf.v[rq].z, kf.v[rq].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          k.re[4 * rq + q] = B::template unpack_lo<DT>(wv[q]);
          k.im[4 * rq + q] = B::template unpack_hi<DT>(wv[q]);
        }
      }
      if constexpr (!SZ) k.im = B::a16_scale(k.im, a.conj_kf ? -1.0f : 1.0f);
      cmul(reA, imA, k); to_op(reA, imA, opA);
      reA = B::a16_zero(); imA = B::a16_zero();
      cmm2<true>(reA, imA, opA, GB);
      cmul(reB, imB, k); to_op(reB, imB, opB);
      reB = B::a16_zero(); imB = B::a16_zero();
      cmm2<true>(reB, imB, opB, GB);
    }
    cmul_conj(reA, imA, R.tw); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm2<true>(reA, imA, opA, GA);
    cmul_conj(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm2<true>(reB, imB, opB, GA);
    tile_store(tau, R, reA, imA);
    tile_store(tau, R, reB, imB, DB);
  }
  template <bool SZ = false>
  static FFC_FN void inner_tile2x(const ConvArgs& a, int h, int tau, const InnerRegs& R, Unit un, uint8_t* zsA = nullptr, uint8_t* zsB = nullptr) {
    static_assert(GEO::N3 == GEO::N2 && GEO::OUTER && GEO::UPW >= 2, "inner_tile2x: two units per workgroup");
    constexpr int DB = GEO::EBYTES;
    KfRegs kf;
    load_kf(a, h, tau, kf);
    Op opA, opB;
    load_tile_op(tau, opA, un, R);
    load_tile_op(tau, opB, un, R, DB);
    A16 reA, imA, reB, imB;
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<false, true>(reA, imA, opA, R.F2);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<false, true>(reB, imB, opB, R.F2);
    cmul(reA, imA, R.tw); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<false, false>(reA, imA, opA, R.F2);
    cmul(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<false, false>(reB, imB, opB, R.F2);
    if constexpr (SZ) { z_store(zsA, tau, reA, imA, FFC_Z_STREAM); z_store(zsB, tau, reB, imB, FFC_Z_STREAM); }     // spectra kept for the backward pass
    // (x) k_f: unpacked once
    {
      CT16 k;
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        u32 wv[4] = {kf.v[rq].x, kf.v[rq].y, kf.v[rq].z, kf.v[rq].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          k.re[4 * rq + q] = B::template unpack_lo<DT>(wv[q]);
          k.im[4 * rq + q] = B::template unpack_hi<DT>(wv[q]);
        }
      }
      if constexpr (!SZ) k.im = B::a16_scale(k.im, a.conj_kf ? -1.0f : 1.0f);
      cmul(reA, imA, k); to_op(reA, imA, opA);
      reA = B::a16_zero(); imA = B::a16_zero();
      cmm<true, true>(reA, imA, opA, R.F2);
      cmul(reB, imB, k); to_op(reB, imB, opB);
      reB = B::a16_zero(); imB = B::a16_zero();
      cmm<true, true>(reB, imB, opB, R.F2);
    }
    cmul_conj(reA, imA, R.tw); to_op(reA, imA, opA);
    reA = B::a16_zero(); imA = B::a16_zero();
    cmm<true, true>(reA, imA, opA, R.F2);
    cmul_conj(reB, imB, R.tw); to_op(reB, imB, opB);
    reB = B::a16_zero(); imB = B::a16_zero();
    cmm<true, true>(reB, imB, opB, R.F2);
    // outer inverse twiddle: one chain per accumulator half, applied to both units' tiles
    {
      const i32 lane = B::opaque(B::lane());
      const i32 c = lane & 31, hi = lane >> 5;
      const i32 sUl = c / GEO::N2, mlane = (c % GEO::N2) * GEO::N3;
      if constexpr (GEO::N3 == 32 && FFC_CHAIN16 != 0) {      // fft 16384: the second half is the first one times w^16 (see twiddle16)
        i32 k1 = sUl * GEO::SV + tau * GEO::G;
        F2 tr[4], ti[4];
        f32 c8, s8;
        chain8p(B::mul24(mlane + hi * 4, k1), k1, 1.0f, a.s_inv, tr, ti, GEO::N - 1, 1.0f / (float)GEO::N, &c8, &s8);
        apply8(reA, imA, 0, tr, ti);
        apply8(reB, imB, 0, tr, ti);
        const f32 c16 = c8 * c8 - s8 * s8, s16 = (c8 + c8) * s8;
#pragma unroll
        for (int i = 0; i < 4; i++) B::cmulp(tr[i], ti[i], c16, s16, tr[i], ti[i]);
        apply8(reA, imA, 1, tr, ti);
        apply8(reB, imB, 1, tr, ti);
      } else
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int sV = (16 * half) / GEO::N3;
        i32 k1 = sUl * GEO::SV + (sV + tau * GEO::G);
        i32 n30 = hi * 4 + ((16 * half) % GEO::N3);
        F2 tr[4], ti[4];
        chain8p(B::mul24(mlane + n30, k1), k1, 1.0f, a.s_inv, tr, ti);
        apply8(reA, imA, half, tr, ti);
        apply8(reB, imB, half, tr, ti);
      }
    }
    tile_store(tau, R, reA, imA);
    tile_store(tau, R, reB, imB, DB);
  }

  // Job loop of the fused sizes.  HALF (32-point outer digit, L <= N/2): only E rows < 16 carry input and
  // only result rows < 16 are stored, so half of the row traffic is skipped and the next pair's rows fit in
  // 32 VGPRs: they are prefetched right after phase A and written to E after rows_out of the current pair.
  // Phase boundary of one unit.  A unit that is handled by a single wave (NW == 1: fft 4096) owns its exchange
  // buffer alone, so its phases only need program order (LDS operations of one wave retire in order): the eight
  // waves of the workgroup are then free to drift apart, one unit's global loads / stores overlap another's math
  // instead of all eight meeting at every phase boundary.
  static FFC_FN void unit_barrier() {
#if defined(FFC_KO) && (FFC_KO & 32)
    B::lds_fence(); return;
#endif
    if constexpr (GEO::NW > 1) B::barrier();
    else B::lds_fence();
  }
  template <bool HALF, bool PROF = false, bool RP = false, bool SZ = false, bool SP = false>
  static FFC_FN void outer_jobs(const ConvArgs& a, int h, int p0, int p1, int u, Unit un, int wg_linear = 0) {
    constexpr int NC = HALF ? NCH / 2 : NCH;
    // HALF: the next pair's rows (32 VGPRs) are prefetched behind the last k_f load of phase B, so they
    // arrive during the last tile / phase C / the stores and no earlier in-order vmcnt wait is delayed.
    // The next pair's rows are requested right before phase C (no other global access until the stores of
    // rows_out) and land in E at the top of the next iteration: their HBM latency hides behind phase C and
    // the stores.  Requesting them any earlier (before phase B) costs more than it gains: the k_f loads of
    // phase B retire in order behind them and the tile loop has no registers to spare
    // (profiles/r01_phase_cycles.txt).
    constexpr bool PREFETCH = HALF && !RP;    // full-length rows: 64 row registers on top of phase C would spill
    // FFC_FOLD_TW: forward / dx kernels of single-pass fft 32768 (not the frequency-sparse, profiling or dynamically scheduled variants)
    constexpr bool FOLDF = CAN_FOLD && !RP && !SP && !PROF && !B::LEAN_OUTER && FFC_FOLD_FWD != 0;
#if defined(FFC_NO_CROSS)
    constexpr bool CROSS = false;
#else
    constexpr bool CROSS = !SP && !RP && !PROF && !B::LEAN_OUTER && GEO::N3 == GEO::N2 && GEO::NW > 1 && GEO::UPW >= 2;
#endif
    // (a split prefetch for the full-length kernels -- first half of the next pair's rows early, second half at the top of
    // the iteration -- was measured: the 32768 kernel then needs 256 VGPRs + 16 spilled and runs the same, 8192 gains 2-4 %;
    // not kept)
    // multi-pass sizes: the R passes of a pair run back to back (pair-major), so that the second read of the input rows and
    // the read-modify-write of the output rows find them in L2
    const int npass = RP ? a.R : 1;
    const int iters = ((p1 - p0 + GEO::UPW - 1) / GEO::UPW) * npass;
    RowRegsT<NC> X;
    if constexpr (PREFETCH) { if (p0 + u < p1) rows_load<NC>(a, h, p0 + u, un, X); }
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
#define FFC_TICK(k) if (PROF) { t1 = B::clock(); acc[k] += t1 - t0; t0 = t1; }
    // Wave priority by progress.  The SIMD issues oldest-first: of its two resident waves (w and w + 4 of the workgroup) the
    // first-dispatched one runs every phase at its single-wave speed, the other gets the issue slots left over (42 % of them
    // in phase B) and then finishes alone -- at the single-wave rate, half the SIMD's (profiles/r03_wave_priority.txt: phase B
    // 12.7 K cycles for waves 0-3, 20.1 K for waves 4-7, which the barrier waits for).  s_setprio steers that arbitration
    // completely (same file: priority 1 on waves 4-7 mirrors the picture), so a wave lowers its priority as it advances through
    // the slices between two barriers: the wave that is a slice behind outranks the wave that is ahead.  Slices are whole
    // phases / phase-B iterations ON PURPOSE: half-iteration slices keep the two waves in the same stage of the same tile, where
    // they compete for the same unit (MFMA against MFMA, twiddle VALU against twiddle VALU), and the kernel ran 12 % slower.
    const bool second = B::wave() >= 4;
    // (fft 4096, one wave per unit and no barriers between the waves: +3 % with priorities, so only where waves share a unit;
    // same-box A/B: forward -2.1 % at fft 32768, -1.2 % at 16384, -2 % at 8192; -DFFC_NO_PRIO builds the kernels without)
#if defined(FFC_NO_PRIO)
#define FFC_PRIO(x)
#else
#define FFC_PRIO(x) if constexpr (GEO::NW > 1) B::template setprio<x>();
#endif
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
      const int p = p0 + (RP ? it / npass : it) * GEO::UPW + u;
      const Pass ps = RP ? make_pass(a, it % npass) : Pass();
      const int hk = RP ? h * ps.R + ps.k0 : h;      // k_f row of this (head, pass)
      const bool act = p < p1;
      if (PROF) t0 = B::clock();
      FFC_PRIO(1)
      if (act) {
        if constexpr (RP) {
          rows_in_rp<NC>(a, h, p, un, ps);
        } else {
          if constexpr (!PREFETCH) rows_load<NC>(a, h, p, un, X);
          rows_store<NC>(a, h, p, un, X);
        }
        B::lds_fence();
        FFC_TICK(0)
        FFC_PRIO(0)
        outer_stage<true, HALF, RP, false, FOLDF>(a.L, un, a.s_fwd, ps);
        FFC_TICK(1)
      }
      unit_barrier();
      FFC_TICK(2)
      if constexpr (CROSS) {
        // cross-unit phase B (inner_tile2x): units 2g and 2g+1 of the workgroup form a group, its 2 NW waves take two tiles
        // each -- of BOTH units when the partner is active, of the first unit alone at the ragged end of a chunk
        const int ug = u >> 1, wl = (u & 1) * GEO::NW + un.wq;
        const int pA = p0 + it * GEO::UPW + 2 * ug;
        if (pA < p1) {
          Unit ua;
          ua.eb = 2 * ug * GEO::EBYTES; ua.wq = wl;
          InnerRegs R;
          load_inner(R, ua);
          uint8_t* zA = SZ ? z_slot(a.zsave, h, a.npair, pA) : nullptr;
          if (pA + 1 < p1) {
#pragma unroll 1
            for (int tt = 0; tt < 2; tt++) {
              if (tt == 0) { FFC_PRIO(3) } else if (second) { FFC_PRIO(2) } else { FFC_PRIO(1) }
              if constexpr (FOLDF) inner_tile2x_fold<SZ>(a, hk, wl * 2 + tt, R, ua, zA, zA + (int64_t)GEO::N * 4);
              else inner_tile2x<SZ>(a, hk, wl * 2 + tt, R, ua, zA, zA + (int64_t)GEO::N * 4);
            }
          } else {
            inner_tile2<false, SZ, FOLDF, HALF>(a, hk, wl * 2, R, ua, Pass(), zA);
          }
        }
      } else if (act) {
        // no long-latency global load may be outstanding while a phase runs: vmcnt retires in order, so
        // anything the compiler spills would wait behind it.  k_f is prefetched one tile ahead only here.
        InnerRegs R;
        load_inner(R, un);
#if defined(FFC_KO) && (FFC_KO & 16)
        if (a.L < 0)
#endif
        if constexpr (SP) {
          Mat Fs;
          lds_mat_sp(Fs);
#pragma unroll 1
          for (int tt = 0; tt < GEO::TPW; tt += 2) inner_tile2_sp(a, hk, un.wq * GEO::TPW + tt, R, Fs, un);
        } else if constexpr (GEO::N3 == GEO::N2) {
          // (a dynamically scheduled phase B -- tiles handed out through an LDS counter -- was measured in round 4, +3 % / -3.5 %, and removed in
          // round 6: profiles/r04_ab_dyn_tiles.txt)
#pragma unroll 1
          for (int tt = 0; tt < GEO::TPW; tt += 2) {
            if (tt == 0) { FFC_PRIO(3) } else if (second) { FFC_PRIO(2) } else { FFC_PRIO(1) }
            inner_tile2<RP, SZ, FOLDF, HALF>(a, hk, un.wq * GEO::TPW + tt, R, un, ps,
                                SZ ? (RP ? z_slot_rp(a.zsave, h, a.npair, p, ps.R, ps.k0) : z_slot(a.zsave, h, a.npair, p)) : nullptr);
          }
        } else {
          KfRegs kf0;
          load_kf(a, h, un.wq * GEO::TPW, kf0);
#pragma unroll 1
          for (int tt = 0; tt < GEO::TPW; tt++) {
            KfRegs kfn;
            if (tt + 1 < GEO::TPW) load_kf(a, h, un.wq * GEO::TPW + tt + 1, kfn);
            inner_tile(a, un.wq * GEO::TPW + tt, R, un, kf0);
            kf0 = kfn;
          }
        }
      }
      FFC_TICK(3)
      unit_barrier();
      FFC_TICK(4)
      FFC_PRIO(3)
      if constexpr (PREFETCH) { if (it + 1 < iters && p + GEO::UPW < p1) rows_load<NC>(a, h, p + GEO::UPW, un, X); }
      if (act) {
        outer_stage<false, HALF, RP>(a.L, un, 1.0f, ps);
        B::lds_fence();
        FFC_TICK(5)
        FFC_PRIO(2)
        if constexpr (SZ) {
          if (a.yraw) {
            ConvArgs ar = a;
            ar.y = a.yraw; ar.postgate = nullptr; ar.sby = (int64_t)a.H * a.L;
            if constexpr (RP) rows_out_rp<NC>(ar, h, p, un, ps);
            else rows_out<NC>(ar, h, p, un);
          }
        }
        if constexpr (RP) rows_out_rp<NC>(a, h, p, un, ps);
        else rows_out<NC>(a, h, p, un);
        FFC_TICK(6)
      }
    }
#undef FFC_TICK
#undef FFC_PRIO
    if (PROF && a.prof) {
      const i32 lane = B::lane();
      unsigned long long* dst = a.prof + ((long long)wg_linear * GEO::WGW + B::wave()) * 8;
#pragma unroll
      for (int k = 0; k < 8; k++) B::g_w64(dst, lane * 0 + k, B::u2_from64(acc[k]), lane < 1);
    }
  }

  // ------------------------------------------------------------------ workgroup entry: conv
  // Workgroup handles head h and one chunk of that head's pairs, UPW units at a time.
  // HALF is chosen by the launcher: 32-point outer digit and L <= N/2
  template <bool HALF = false, bool SZ = false, bool SP = false>
  static FFC_FN void conv(const ConvArgs& a, int h, int chunk) {
    setup_tables(a.tab, a.t);
    conv_job<HALF, false, SZ, SP>(a, h, chunk);
  }
  // one (head, chunk) job; the tables are already in LDS.  RP: all passes of a multi-pass size
  template <bool HALF = false, bool RP = false, bool SZ = false, bool SP = false>
  static FFC_FN void conv_job(const ConvArgs& a, int h, int chunk) {
    const int wv = B::wave();
    Unit un;
    un.wq = wv % GEO::NW;
    const int u = wv / GEO::NW;
    un.eb = u * GEO::EBYTES;
    const int p0 = chunk * a.ppc;
    int p1 = p0 + a.ppc;
    if (p1 > a.npair) p1 = a.npair;
    if constexpr (GEO::OUTER) {
      outer_jobs<HALF, false, RP, SZ, SP>(a, h, p0, p1, u, un);
    } else {
      // one tile of G pairs per wave
      const int q0 = p0 / GEO::G, q1 = (p1 + GEO::G - 1) / GEO::G;
      const int iters = (q1 - q0 + GEO::UPW - 1) / GEO::UPW;
      InnerRegs R;
      load_inner(R, un);
      if constexpr (RP) {
        // multi-pass inner-only form (fft 2048 on the 32 x 32 kernel): G == 1, the R passes of a pair back to back
        static_assert(GEO::G == 1, "inner-only multi-pass: one pair per tile");
        // Round 6: rows that fit one block (L <= 1024: the padded case, sum_n0 has one term) -- the pair's rows are loaded ONCE and stay
        // in registers for every pass, and the passes' outputs are summed in registers and stored ONCE.  Before, every pass loaded the
        // rows again and passes k0 > 0 read-modify-wrote the output rows (and the pre-postgate copy of the gated training forward):
        // per pair 2 loads of u, 2 stores + 1 load of y instead of one each -- a third of the launch's traffic at fft 2048, whose
        // kernels run at the memory system's rate (profiles/r06_ab_small_pipe.txt).  Same arithmetic in the same order: bit-identical
        // (FFC_IP_MERGE=0: the per-pass form, tests/test_build_switches.py).
#ifndef FFC_IP_MERGE
#define FFC_IP_MERGE 1
#endif
        if (FFC_IP_MERGE != 0 && a.fast && a.L <= GEO::N && !a.aux_in) {
          const i32 lane = B::opaque(B::lane());
#pragma unroll 1
          for (int it = 0; it < iters; it++) {
            const int q = q0 + it * GEO::UPW + u;
            if (q >= q1) continue;
            RowRegs X, Y;
            rows_load<NCH>(a, h, q, un, X);
            if (a.pregate) {
              RowRegs G;
#pragma unroll
              for (int ii = 0; ii < NCH; ii++) {
                i32 idx = lane + ii * 64;
                i32 m = (idx % CPR) * 8;
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                  i32 b = (idx / CPR + q * GEO::G) * 2 + pl;
                  G.v[ii][pl] = gload8_rows((const uint16_t*)a.pregate, b, h, a, a.sbg, m, 1, b < a.B);
                }
              }
#pragma unroll
              for (int ii = 0; ii < NCH; ii++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) X.v[ii][pl] = mul4(X.v[ii][pl], G.v[ii][pl]);
            }
#pragma unroll 1
            for (int k0 = 0; k0 < a.R; k0++) {
              InnerPass ip;
              load_inner_pass(ip, k0);
              KfRegs kf;
              load_kf(a, h * a.R + k0, 0, kf);
#pragma unroll
              for (int ii = 0; ii < NCH; ii++) {
                i32 idx = lane + ii * 64;
                pred sw;
                i32 off = pair_off(idx / CPR, (idx % CPR) * 8, &sw) + un.eb;
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                  const U4& v = X.v[ii][pl];
                  U4 o;
                  o.x = B::sel(sw, v.z, v.x); o.y = B::sel(sw, v.w, v.y);
                  o.z = B::sel(sw, v.x, v.z); o.w = B::sel(sw, v.y, v.w);
                  B::lds_w128(off + pl * GEO::PLANE, o, B::ptrue());
                }
              }
              B::lds_fence();
              inner_tile<true, SZ>(a, 0, R, un, kf, &ip, (SZ && a.zsave) ? z_slot_small(a.zsave, h, a.npair, q, a.R, k0) : nullptr);
              B::lds_fence();
#pragma unroll
              for (int ii = 0; ii < NCH; ii++) {
                i32 idx = lane + ii * 64;
                pred sw;
                i32 off = pair_off(idx / CPR, (idx % CPR) * 8, &sw) + un.eb;
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                  U4 o = B::lds_r128(off + pl * GEO::PLANE);
                  U4 v;
                  v.x = B::sel(sw, o.z, o.x); v.y = B::sel(sw, o.w, o.y);
                  v.z = B::sel(sw, o.x, o.z); v.w = B::sel(sw, o.y, o.w);
                  // y[m] (+)= y_k0[m] (n0 = 0: no rotation), fp32 sum rounded once, as rows_out_rp
                  if (k0 == 0) Y.v[ii][pl] = v;
                  else Y.v[ii][pl] = add4(Y.v[ii][pl], v, 1.0f);
                }
              }
              B::lds_fence();
            }
#pragma unroll
            for (int ii = 0; ii < NCH; ii++) {
              i32 idx = lane + ii * 64;
              i32 m = (idx % CPR) * 8;
#pragma unroll
              for (int pl = 0; pl < 2; pl++) {
                i32 b = (idx / CPR + q * GEO::G) * 2 + pl;
                U4 v = Y.v[ii][pl];
                if constexpr (SZ) {
                  if (a.yraw) gstore8_rows((uint16_t*)a.yraw, b, h, a, (int64_t)a.H * a.L, m, 1, b < a.B, v);      // output before the postgate
                }
                if (a.postgate) v = mul4(v, gload8_rows((const uint16_t*)a.postgate, b, h, a, a.sbp, m, 1, b < a.B));
                gstore8_rows((uint16_t*)a.y, b, h, a, a.sby, m, 1, b < a.B, v);
              }
            }
          }
          return;
        }
#pragma unroll 1
        for (int it = 0; it < iters * a.R; it++) {
          const int q = q0 + (it / a.R) * GEO::UPW + u;
          const int k0 = it % a.R;
          if (q < q1) {
            Pass ps; ps.k0 = k0; ps.R = a.R;
            InnerPass ip;
            load_inner_pass(ip, k0);
            KfRegs kf;
            load_kf(a, h * a.R + k0, 0, kf);
            rows_in_rp<NCH>(a, h, q, un, ps);
            B::lds_fence();
            inner_tile<true, SZ>(a, 0, R, un, kf, &ip, (SZ && a.zsave) ? z_slot_small(a.zsave, h, a.npair, q, a.R, k0) : nullptr);
            B::lds_fence();
            if constexpr (SZ) {
              if (a.yraw) {      // output before the postgate multiply (dpostgate = dout * this)
                ConvArgs ar = a;
                ar.y = a.yraw; ar.postgate = nullptr; ar.sby = (int64_t)a.H * a.L;
                rows_out_rp<NCH>(ar, h, q, un, ps);
              }
            }
            rows_out_rp<NCH>(a, h, q, un, ps);
            B::lds_fence();
          }
        }
        return;
      }
#pragma unroll 1
      for (int it = 0; it < iters; it++) {
        const int q = q0 + it * GEO::UPW + u;
        const bool act = q < q1;
        if (act) {
          KfRegs kf;
          load_kf(a, h, 0, kf);
          rows_in(a, h, q, un);
          B::lds_fence();
          inner_tile<false, SZ>(a, 0, R, un, kf, nullptr, (SZ && a.zsave) ? z_slot_small(a.zsave, h, a.npair, q, 1, 0) : nullptr);
          B::lds_fence();
          if constexpr (SZ) {
            if (a.yraw) {
              ConvArgs ar = a;
              ar.y = a.yraw; ar.postgate = nullptr; ar.sby = (int64_t)a.H * a.L;
              rows_out(ar, h, q, un);
            }
          }
          rows_out(a, h, q, un);
        }
      }
    }
  }
};

}  // namespace ffc
