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

#ifndef FFC_FN
#define FFC_FN inline __attribute__((always_inline))
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
};

template <class B, class GEO, int DT>
struct Body {
  using f32 = typename B::f32;
  using i32 = typename B::i32;
  using u32 = typename B::u32;
  using pred = typename B::pred;
  using U2 = typename B::U2;
  using U4 = typename B::U4;

  struct Mat { u32 w[2][3][4]; };   // [K-step][Fr,Fi,-Fi][dword]
  struct CT16 { f32 re[16], im[16]; };
  struct Op { u32 r[2][4], i[2][4]; };  // complex MFMA data operand, 2 K-steps

  static FFC_FN void load_mat(Mat& m, const uint8_t* p, i32 lane) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
      U4 v = B::g_r128(p, lane + q * 64);
      m.w[q / 3][q % 3][0] = v.x; m.w[q / 3][q % 3][1] = v.y;
      m.w[q / 3][q % 3][2] = v.z; m.w[q / 3][q % 3][3] = v.w;
    }
  }
  static FFC_FN void load_ct16(CT16& c, const uint8_t* p, i32 lane) {
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      U4 v = B::g_r128(p, lane + rr * 64);
      c.re[2 * rr] = B::as_f32(v.x); c.im[2 * rr] = B::as_f32(v.y);
      c.re[2 * rr + 1] = B::as_f32(v.z); c.im[2 * rr + 1] = B::as_f32(v.w);
    }
  }
  static FFC_FN void zero(f32 (&a)[16]) {
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = B::fconst(0.f);
  }
  // acc += M (x) data.  AFORM: data is the MFMA A operand (its lane index moves to registers,
  // the transformed index lands on lanes).  !AFORM: matrix is A (transformed index in registers,
  // data lane index stays on lanes).  CONJ selects the inverse DFT.  ms_lim: K-steps to run.
  template <bool CONJ, bool AFORM>
  static FFC_FN void cmm(f32 (&ore)[16], f32 (&oim)[16], const Op& d, const Mat& F, int ms_lim = 2) {
#pragma unroll
    for (int ms = 0; ms < 2; ms++) {
      if (ms >= ms_lim) break;
      const u32(&fr)[4] = F.w[ms][0];
      const u32(&fi_re)[4] = F.w[ms][CONJ ? 1 : 2];  // multiplies data.im into re
      const u32(&fi_im)[4] = F.w[ms][CONJ ? 2 : 1];  // multiplies data.re into im
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
  static FFC_FN void to_op(const f32 (&re)[16], const f32 (&im)[16], Op& o) {
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        o.r[ms][d] = B::template pack<DT>(re[8 * ms + 2 * d], re[8 * ms + 2 * d + 1]);
        o.i[ms][d] = B::template pack<DT>(im[8 * ms + 2 * d], im[8 * ms + 2 * d + 1]);
      }
  }
  static FFC_FN void cmul(f32 (&re)[16], f32 (&im)[16], const CT16& t) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      f32 a = re[r], b = im[r];
      re[r] = a * t.re[r] - b * t.im[r];
      im[r] = a * t.im[r] + b * t.re[r];
    }
  }
  // dtype pair (x) dtype pair, rounded back to dtype (the reference multiplies gates in the
  // activation dtype: kernels_bf16/monarch_cuda_32_32_32_kernel_bf16.h:409-429, 613-634).
  static FFC_FN u32 mul2(u32 a, u32 g) {
    f32 lo = B::template unpack_lo<DT>(a) * B::template unpack_lo<DT>(g);
    f32 hi = B::template unpack_hi<DT>(a) * B::template unpack_hi<DT>(g);
    return B::template pack<DT>(lo, hi);
  }
  // raw[e] holds 4 consecutive columns (tiles t=0..3) of k-slot e; build tile t's operand dwords.
  static FFC_FN void xpose(const U2 (&raw)[8], int t, u32 (&op)[4]) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
      u32 a = (t < 2) ? raw[2 * d].x : raw[2 * d].y;
      u32 b = (t < 2) ? raw[2 * d + 1].x : raw[2 * d + 1].y;
      op[d] = (t & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
  }

  // ------------------------------------------------------------------ phase A (outer fwd)
  static FFC_FN void phase_a(const ConvArgs& a, int h, int p, const Mat& F1, const CT16& base) {
    const i32 lane = B::lane();
    const int w = B::wave();
    const i32 j = lane & 31, hi = lane >> 5;
    const int b0 = 2 * p, b1 = 2 * p + 1;
    const bool v1 = b1 < a.B;
    const int64_t rowa = ((int64_t)b0 * a.H + h) * a.L, rowb = ((int64_t)(v1 ? b1 : b0) * a.H + h) * a.L;
    const uint16_t* xa = (const uint16_t*)a.u + rowa;
    const uint16_t* xb = (const uint16_t*)a.u + rowb;
    const uint16_t* ga = a.pregate ? (const uint16_t*)a.pregate + rowa : nullptr;
    const uint16_t* gb = a.pregate ? (const uint16_t*)a.pregate + rowb : nullptr;
    int ms_lim = 2;
    if (GEO::S1 == 1 && 16 * GEO::Mi >= a.L) ms_lim = 1;

    U2 rawa[2][8], rawb[2][8];
#pragma unroll
    for (int ms = 0; ms < 2; ms++) {
      if (ms >= ms_lim) break;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        i32 R = hi * 4 + (16 * ms + 8 * (e >> 2) + (e & 3));
        i32 s1 = R / GEO::N1, n1 = R % GEO::N1;
        i32 n = n1 * GEO::Mi + s1 * 128 + j * 4 + w * 128 * GEO::S1;
        pred ok = n < a.L;
        i32 o8 = n >> 2;
        rawa[ms][e] = B::g_r64(xa, o8, ok);
        rawb[ms][e] = B::g_r64(xb, o8, v1 ? ok : B::pfalse());
        if (ga) {
          U2 g = B::g_r64(ga, o8, ok);
          rawa[ms][e].x = mul2(rawa[ms][e].x, g.x); rawa[ms][e].y = mul2(rawa[ms][e].y, g.y);
          U2 g2 = B::g_r64(gb, o8, v1 ? ok : B::pfalse());
          rawb[ms][e].x = mul2(rawb[ms][e].x, g2.x); rawb[ms][e].y = mul2(rawb[ms][e].y, g2.y);
        }
      }
    }
    u32 sre[16][2], sim[16][2];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      Op op;
#pragma unroll
      for (int ms = 0; ms < 2; ms++) {
        if (ms >= ms_lim) break;
        xpose(rawa[ms], t, op.r[ms]);
        xpose(rawb[ms], t, op.i[ms]);
      }
      f32 re[16], im[16];
      zero(re); zero(im);
      cmm<false, false>(re, im, op, F1, ms_lim);
      if (t == 0) {
        cmul(re, im, base);
      } else {
        CT16 c;
        load_ct16(c, a.tab + a.t.ct + 8192 * t, lane);
        cmul(c.re, c.im, base);
        cmul(re, im, c);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        u32 vr = B::template pack<DT>(re[r], B::fconst(0.f));
        u32 vi = B::template pack<DT>(im[r], B::fconst(0.f));
        if (t & 1) {
          sre[r][t >> 1] = sre[r][t >> 1] | (vr << 16);
          sim[r][t >> 1] = sim[r][t >> 1] | (vi << 16);
        } else {
          sre[r][t >> 1] = vr & 0xffffu;
          sim[r][t >> 1] = vi & 0xffffu;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      i32 R = hi * 4 + ((r & 3) + 8 * (r >> 2));
      i32 s1 = R / GEO::N1, k1 = R % GEO::N1;
      i32 m = s1 * 128 + j * 4 + w * 128 * GEO::S1;
      i32 off = e_off<GEO, i32>(k1, m);
      U2 vr; vr.x = sre[r][0]; vr.y = sre[r][1];
      U2 vi; vi.x = sim[r][0]; vi.y = sim[r][1];
      B::lds_w64(off, vr);
      B::lds_w64(off + GEO::PLANE, vi);
    }
  }

  // ------------------------------------------------------------------ phase C (outer inverse)
  static FFC_FN void phase_c(const ConvArgs& a, int h, int p, const Mat& F1) {
    const i32 lane = B::lane();
    const int w = B::wave();
    const i32 j = lane & 31, hi = lane >> 5;
    const int b0 = 2 * p, b1 = 2 * p + 1;
    const bool v1 = b1 < a.B;
    const int64_t rowa = ((int64_t)b0 * a.H + h) * a.L, rowb = ((int64_t)(v1 ? b1 : b0) * a.H + h) * a.L;
    uint16_t* ya = (uint16_t*)a.y + rowa;
    uint16_t* yb = (uint16_t*)a.y + rowb;
    const uint16_t* ga = a.postgate ? (const uint16_t*)a.postgate + rowa : nullptr;
    const uint16_t* gb = a.postgate ? (const uint16_t*)a.postgate + rowb : nullptr;

    U2 rawr[2][8], rawi[2][8];
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        i32 R = hi * 4 + (16 * ms + 8 * (e >> 2) + (e & 3));
        i32 s1 = R / GEO::N1, k1 = R % GEO::N1;
        i32 m = s1 * 128 + j * 4 + w * 128 * GEO::S1;
        i32 off = e_off<GEO, i32>(k1, m);
        rawr[ms][e] = B::lds_r64(off);
        rawi[ms][e] = B::lds_r64(off + GEO::PLANE);
      }
    u32 sre[16][2], sim[16][2];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      Op op;
#pragma unroll
      for (int ms = 0; ms < 2; ms++) {
        xpose(rawr[ms], t, op.r[ms]);
        xpose(rawi[ms], t, op.i[ms]);
      }
      f32 re[16], im[16];
      zero(re); zero(im);
      cmm<true, false>(re, im, op, F1);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        u32 vr = B::template pack<DT>(re[r], B::fconst(0.f));
        u32 vi = B::template pack<DT>(im[r], B::fconst(0.f));
        if (t & 1) {
          sre[r][t >> 1] = sre[r][t >> 1] | (vr << 16);
          sim[r][t >> 1] = sim[r][t >> 1] | (vi << 16);
        } else {
          sre[r][t >> 1] = vr & 0xffffu;
          sim[r][t >> 1] = vi & 0xffffu;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      i32 R = hi * 4 + ((r & 3) + 8 * (r >> 2));
      i32 s1 = R / GEO::N1, n1 = R % GEO::N1;
      i32 n = n1 * GEO::Mi + s1 * 128 + j * 4 + w * 128 * GEO::S1;
      pred ok = n < a.L;
      i32 o8 = n >> 2;
      U2 vr; vr.x = sre[r][0]; vr.y = sre[r][1];
      U2 vi; vi.x = sim[r][0]; vi.y = sim[r][1];
      if (ga) {
        U2 g = B::g_r64(ga, o8, ok);
        vr.x = mul2(vr.x, g.x); vr.y = mul2(vr.y, g.y);
        U2 g2 = B::g_r64(gb, o8, v1 ? ok : B::pfalse());
        vi.x = mul2(vi.x, g2.x); vi.y = mul2(vi.y, g2.y);
      }
      B::g_w64(ya, o8, vr, ok);
      B::g_w64(yb, o8, vi, v1 ? ok : B::pfalse());
    }
  }

  // ------------------------------------------------------------------ copy in / out (N <= 1024)
  // Tile of G pairs (pairs q*G .. q*G+G-1 of head h); E row g = pair g, re = row 2p, im = row 2p+1.
  static FFC_FN void copy_in(const ConvArgs& a, int h, int q) {
    const i32 lane = B::lane();
    const uint16_t* ub = (const uint16_t*)a.u + (int64_t)h * a.L;
    const uint16_t* gbse = a.pregate ? (const uint16_t*)a.pregate + (int64_t)h * a.L : nullptr;
    const int64_t bstride8 = ((int64_t)a.H * a.L) >> 2;   // 8-byte units per batch row
#pragma unroll
    for (int i = 0; i < 4; i++) {
      i32 ci = lane + i * 64;                 // chunk id within a plane (256 chunks of 4)
      i32 g = ci / (GEO::Mi / 4), m = (ci % (GEO::Mi / 4)) * 4;
      i32 pp = g + q * GEO::G;                // pair index
      i32 bA = pp * 2, bB = pp * 2 + 1;
      pred inl = m < a.L;
      pred okA = inl && (bA < a.B), okB = inl && (bB < a.B);
      i32 oA = bA * (int)bstride8 + (m >> 2), oB = bB * (int)bstride8 + (m >> 2);
      U2 va = B::g_r64(ub, oA, okA), vb = B::g_r64(ub, oB, okB);
      if (gbse) {
        U2 g1 = B::g_r64(gbse, oA, okA), g2 = B::g_r64(gbse, oB, okB);
        va.x = mul2(va.x, g1.x); va.y = mul2(va.y, g1.y);
        vb.x = mul2(vb.x, g2.x); vb.y = mul2(vb.y, g2.y);
      }
      i32 off = e_off<GEO, i32>(g, m);
      B::lds_w64(off, va);
      B::lds_w64(off + GEO::PLANE, vb);
    }
  }
  static FFC_FN void copy_out(const ConvArgs& a, int h, int q) {
    const i32 lane = B::lane();
    uint16_t* yb = (uint16_t*)a.y + (int64_t)h * a.L;
    const uint16_t* gbse = a.postgate ? (const uint16_t*)a.postgate + (int64_t)h * a.L : nullptr;
    const int64_t bstride8 = ((int64_t)a.H * a.L) >> 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      i32 ci = lane + i * 64;
      i32 g = ci / (GEO::Mi / 4), m = (ci % (GEO::Mi / 4)) * 4;
      i32 pp = g + q * GEO::G;
      i32 bA = pp * 2, bB = pp * 2 + 1;
      pred inl = m < a.L;
      pred okA = inl && (bA < a.B), okB = inl && (bB < a.B);
      i32 oA = bA * (int)bstride8 + (m >> 2), oB = bB * (int)bstride8 + (m >> 2);
      i32 off = e_off<GEO, i32>(g, m);
      U2 va = B::lds_r64(off), vb = B::lds_r64(off + GEO::PLANE);
      if (gbse) {
        U2 g1 = B::g_r64(gbse, oA, okA), g2 = B::g_r64(gbse, oB, okB);
        va.x = mul2(va.x, g1.x); va.y = mul2(va.y, g1.y);
        vb.x = mul2(vb.x, g2.x); vb.y = mul2(vb.y, g2.y);
      }
      B::g_w64(yb, oA, va, okA);
      B::g_w64(yb, oB, vb, okB);
    }
  }

  // ------------------------------------------------------------------ phase B (inner tile)
  struct InnerRegs { Mat F2, F3; CT16 tw, tw2; };

  static FFC_FN void load_tile_op(int tau, Op& op) {
    const i32 lane = B::lane();
    const i32 c = lane & 31, hi = lane >> 5;
    const i32 sV = c / GEO::N3;
    if (B::HAS_TR) {
      const i32 i16 = lane & 15, g16 = (lane >> 4) & 1;
      const i32 n3b = (g16 * 16) % GEO::N3;
#pragma unroll
      for (int ms = 0; ms < 2; ms++)
#pragma unroll
        for (int rho = 0; rho < 2; rho++) {
          i32 U = hi * 4 + (16 * ms + 8 * rho) + (i16 >> 2);   // this lane supplies row U
          i32 sU = U / GEO::N2, n2 = U % GEO::N2;
          i32 row = sU * GEO::SV + ((g16 * 16) / GEO::N3) + tau * GEO::G;
          i32 m = n2 * GEO::N3 + n3b + (i16 & 3) * 4;
          i32 off = e_off<GEO, i32>(row, m);
          U2 vr = B::lds_r64_tr(off), vi = B::lds_r64_tr(off + GEO::PLANE);
          op.r[ms][2 * rho] = vr.x; op.r[ms][2 * rho + 1] = vr.y;
          op.i[ms][2 * rho] = vi.x; op.i[ms][2 * rho + 1] = vi.y;
        }
    } else {
      const i32 n3 = c % GEO::N3;
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
            i32 off = e_off<GEO, i32>(row, n2 * GEO::N3 + n3);
            wr[hf] = B::lds_r16(off);
            wi[hf] = B::lds_r16(off + GEO::PLANE);
          }
          op.r[ms][d] = wr[0] | (wr[1] << 16);
          op.i[ms][d] = wi[0] | (wi[1] << 16);
        }
    }
  }

  static FFC_FN void inner_tile(const ConvArgs& a, int h, int tau, const InnerRegs& R) {
    const i32 lane = B::lane();
    const i32 c = lane & 31, hi = lane >> 5;
    Op op;
    load_tile_op(tau, op);
    f32 re[16], im[16];
    // stage a: contract n2 (A-form) -> [V=(sV,n3) regs][U'=(sU,k2) lanes]
    zero(re); zero(im);
    cmm<false, true>(re, im, op, R.F2);
    cmul(re, im, R.tw);
    to_op(re, im, op);
    // stage b: contract n3 (B-form) -> [V'=(sV,k3) regs][U' lanes]
    zero(re); zero(im);
    cmm<false, false>(re, im, op, R.F3);
    // (x) k_f
    {
      const uint8_t* kfh = (const uint8_t*)a.kf + (int64_t)h * (GEO::NT * 1024 * 4);
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c);
        U4 v = B::g_r128(kfh, idx);
        u32 wv[4] = {v.x, v.y, v.z, v.w};
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
    }
    to_op(re, im, op);
    // inverse stage b: contract k3 (A-form, conj) -> [U' regs][V''=(sV,n3) lanes]
    zero(re); zero(im);
    cmm<true, true>(re, im, op, R.F3);
    cmul(re, im, R.tw2);
    to_op(re, im, op);
    // inverse stage a: contract k2 (A-form, conj) -> [V'' regs][U''=(sU,n2) lanes]
    zero(re); zero(im);
    cmm<true, true>(re, im, op, R.F2);
    // outer inverse twiddle W_N^{-(n2*N3+n3)*k1}
    if constexpr (GEO::OUTER) {
      const uint8_t* pa = a.tab + a.t.oi_a + (int64_t)tau * (32 * GEO::SV * 8);
      const uint8_t* pb = a.tab + a.t.oi_b + (int64_t)tau * (GEO::SU * 2 * 16 * 8);
      f32 are[GEO::SV], aim[GEO::SV];
#pragma unroll
      for (int s = 0; s < GEO::SV; s++) {
        U2 v = B::g_r64(pa, c * GEO::SV + s, B::ptrue());
        are[s] = B::as_f32(v.x); aim[s] = B::as_f32(v.y);
      }
      i32 bidx = ((c / GEO::N2) * 2 + hi) * 8;   // in 16-byte units (2 complex each)
#pragma unroll
      for (int rr = 0; rr < 8; rr++) {
        U4 v = B::g_r128(pb, bidx + rr);
        f32 br[2] = {B::as_f32(v.x), B::as_f32(v.z)}, bi[2] = {B::as_f32(v.y), B::as_f32(v.w)};
#pragma unroll
        for (int q = 0; q < 2; q++) {
          int r = 2 * rr + q;
          int s = (GEO::SV == 1) ? 0 : (r >> 3);
          f32 tr = are[s] * br[q] - aim[s] * bi[q];
          f32 ti = are[s] * bi[q] + aim[s] * br[q];
          f32 x = re[r], y = im[r];
          re[r] = x * tr - y * ti;
          im[r] = x * ti + y * tr;
        }
      }
    }
    // write back in place: lane <-> (sU,n2), regs <-> (sV,n3); r&3 = 4 consecutive n3
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 V = hi * 4 + 8 * rq;
      i32 sV = V / GEO::N3, n3 = V % GEO::N3;
      i32 sU = c / GEO::N2, n2 = c % GEO::N2;
      i32 row = sU * GEO::SV + sV + tau * GEO::G;
      i32 off = e_off<GEO, i32>(row, n2 * GEO::N3 + n3);
      U2 vr, vi;
      vr.x = B::template pack<DT>(re[4 * rq], re[4 * rq + 1]);
      vr.y = B::template pack<DT>(re[4 * rq + 2], re[4 * rq + 3]);
      vi.x = B::template pack<DT>(im[4 * rq], im[4 * rq + 1]);
      vi.y = B::template pack<DT>(im[4 * rq + 2], im[4 * rq + 3]);
      B::lds_w64(off, vr);
      B::lds_w64(off + GEO::PLANE, vi);
    }
  }

  // ------------------------------------------------------------------ workgroup entry: conv
  // Workgroup wg handles head h and one chunk of that head's pairs.
  static FFC_FN void conv(const ConvArgs& a, int h, int chunk) {
    const i32 lane = B::lane();
    InnerRegs R;
    load_mat(R.F2, a.tab + a.t.mat[1], lane);
    if (GEO::N3 != GEO::N2) load_mat(R.F3, a.tab + a.t.mat[2], lane); else R.F3 = R.F2;
    load_ct16(R.tw, a.tab + a.t.twin, lane);
    load_ct16(R.tw2, a.tab + a.t.twin2, lane);
    const int p0 = chunk * a.ppc;
    int p1 = p0 + a.ppc;
    if (p1 > a.npair) p1 = a.npair;
    if constexpr (GEO::OUTER) {
      const int w = B::wave();
      Mat F1;
      CT16 base;
      load_mat(F1, a.tab + a.t.mat[0], lane);
      load_ct16(base, a.tab + a.t.base + 8192 * w, lane);
      for (int p = p0; p < p1; p++) {
        phase_a(a, h, p, F1, base);
        B::barrier();
        for (int tt = 0; tt < GEO::TPW; tt++) inner_tile(a, h, w * GEO::TPW + tt, R);
        B::barrier();
        phase_c(a, h, p, F1);
        B::barrier();
      }
    } else {
      // tiles of G pairs
      const int q0 = p0 / GEO::G, q1 = (p1 + GEO::G - 1) / GEO::G;
      for (int q = q0; q < q1; q++) {
        copy_in(a, h, q);
        B::barrier();
        inner_tile(a, h, 0, R);
        B::barrier();
        copy_out(a, h, q);
        B::barrier();
      }
    }
  }
};

}  // namespace ffc
