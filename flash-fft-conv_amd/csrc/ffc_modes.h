// Kernel modes built from the conv building blocks (ffc_body.h):
//   kfft  : k (H,Lk) fp32 -> k_f in internal order           (replaces torch.fft.fft + permute + cast,
//           reference flashfftconv/conv.py:572-575, :585, :676)
//   dkf   : W[h] = sum_b FFT(dout*postgate) * conj(FFT(u*pregate)), fp32 partial slabs
//           (reference: the dk_f half of kernels_bf16/*_bwd_kernel_bf16.h, which accumulates in bf16
//           registers and reduces with dk_f_out.sum(0), monarch_cuda_interface_bwd_bf16.cu:1256-1264)
//   dkifft: dk = Re(iFFT(sum of slabs))[:Lk]                   (reference conv.py:1758-1761, 1861-1864)
// Pair packing keeps working for dk: with z_u = u_a + i u_b and z_d = d_a + i d_b,
//   Re iFFT( Z_d * conj(Z_u) ) = corr(d_a,u_a) + corr(d_b,u_b).
#pragma once
#ifndef FFC_IP_LEAN
#define FFC_IP_LEAN 1      // 0: the pass's tables resident in 80 registers (round 5: 34 - 43 registers spilled; A/B builds)
#endif
#include "ffc_body.h"

namespace ffc {

struct KfArgs {
  const float* k;      // (H, Lk) fp32
  void* kf;            // (H, NT*1024, 2) dtype, internal order
  const uint8_t* tab;
  PlanTabs t;
  int H, Lk;
  float scale;         // s_k / (s_fwd * prescale)
  float s_fwd;         // plan forward scale (outer forward twiddle)
  float prescale;      // applied to k before rounding to dtype (2^8 in fp16 mode: k's energy sits in a few taps,
                       // its scaled spectrum would otherwise fall into the fp16 subnormal range)
  const void* xpair;   // optional complex input instead of k: pair-plane tensor (2, H, M) dtype (big FFT sizes)
  int fast;            // Lk % 4 == 0 and 16-byte aligned
  int R;               // > 1: multi-pass size (struct Pass); k_f rows are (head, pass)
};

struct DkfArgs {
  ConvArgs c;          // u / pregate / postgate as in the forward; c.y unused
  const void* dout;
  float* ws;           // [nchunk*UPW][H][NT*1024][2] fp32 partial sums (internal order)
  int64_t sbd, sbdu, sbdpre, sbdpost;   // batch strides (elements) of dout / du / dpre / dpost (see ConvArgs::sbu)
  // fused backward (Modes::bwd) only: c.kf = k_f, du = pregate * corr(dout*postgate, k),
  // dpre = u * corr(dout*postgate, k) (nullable)
  void* du;
  void* dpre;
  // dpost = dout * conv(u*pregate, k) (nullable; fused sizes >= 4096 only): the forward output falls out of the
  // first spectrum of each pair (one extra inverse transform instead of a second forward launch)
  void* dpost;
  // scratch for the first spectrum of a pair (dtype, internal order), one N-point slot per (workgroup, unit):
  // written and read back by the same wave, so it only has to survive in L2 (fused sizes >= 4096)
  void* zscratch;
  // optional (ffc_conv_bwd_z): the spectra FFT(u * pregate) saved by the forward pass (ConvArgs::zsave layout).  The kernel
  // then skips its first transform of every pair (rows of u, outer stage, two inner stages, scratch round trip).
  const void* zin;
  // with zin, optional (ffc_conv_bwd_zy): the forward output before the postgate multiply, contiguous (B,H,L) dtype, as stored by
  // ffc_conv_fwd_z.  dpost = dout * yraw is then a side product of the dout row load (ConvArgs::aux_in) instead of an inverse
  // transform of the saved spectrum.
  const void* yraw;
  // optional (round 4; fused backward only): dk (H, Lk) fp32 written by the SAME launch.  When a workgroup owns every pair of its
  // head (nchunk == 1) its accumulation registers hold the head's whole dk_f at the end of the pair loop, so the inverse
  // transform of Modes::dkifft can run right there, from the registers: no 256 KB fp32 slab per head written and read back, no
  // second launch (config 2: 201 MB each way and 0.07 ms of a 1.28 ms step).  Single-pass fft 32768, bf16 plans (the dk
  // inverse always runs on bf16 tables: the kernel's own); the launcher falls back to slab + dkifft otherwise.
  float* dk_out;
  int Lk, dk_fast;
  float dk_scale;      // 1 / s_fwd (W carries s_fwd^2, tile_inv applies 1 / (N s_fwd))
  // fp16 plans: the plan's bf16 tables (the dk inverse runs in bf16 operand arithmetic for fp32's range, as ffc_kernel_ifft_grad
  // does).  The fp16 kernel swaps them into LDS behind its pair loop and runs the bf16 instantiation of the tail.
  const uint8_t* tab_bf;
  PlanTabs t_bf;
  // instead of dk_out: complex output of the tail, pair-plane tensor (2, H, N) bf16 (first step of dk at the HBM-level sizes, as
  // ffc_kernel_ifft_grad_c); dk_scale is then the caller's scale
  void* dk_pair;
};

struct DkArgs {
  const float* ws;
  float* dk;           // (H, Lk) fp32
  const uint8_t* tab;
  PlanTabs t;
  int H, Lk, nslab;
  float scale;         // 1 / s_fwd  (W carries s_fwd^2, the inverse applies 1/(N s_fwd))
  float s_inv;         // plan's 1/(N*s_fwd) (applied inside tile_inv for fused sizes >= 4096)
  int flags;           // debugging switches
  void* outpair;       // optional complex output instead of dk: pair-plane tensor (2, H, M) dtype (big FFT sizes)
  int fast;
  int R;               // > 1: multi-pass size (struct Pass); slab rows are (head, pass)
};

#ifndef FFC_WREG
#define FFC_WREG 4
#endif
template <class B, class GEO, int DT>
struct Modes : Body<B, GEO, DT> {
  using BD = Body<B, GEO, DT>;
  using f32 = typename B::f32;
  using i32 = typename B::i32;
  using u32 = typename B::u32;
  using pred = typename B::pred;
  using U2 = typename B::U2;
  using U4 = typename B::U4;
  using A16 = typename B::A16;
  using W4 = typename B::W4;
  using Unit = typename BD::Unit;
  using Op = typename BD::Op;
  using InnerRegs = typename BD::InnerRegs;
  using InnerPass = typename BD::InnerPass;
  static FFC_FN Pass make_pass(const uint8_t* tab, const PlanTabs& t, int R, int k0) {
    Pass ps;
    ps.k0 = k0; ps.R = R; ps.mat_fwd = tab + t.matk[k0][0]; ps.mat_inv = tab + t.matk[k0][1];
    return ps;
  }

  // ------------------------------------------------------------------ k -> k_f
  // 8 fp32 starting at element e0 (per lane) of base, zero beyond `lim` (per lane limit on e0+i)
  static FFC_FN void fload8(const float* base, i32 e0, i32 n, int lim, bool fast, pred ok, f32 (&v)[8]) {
    if (fast) {
      U4 a = B::g_r128p(base, e0 >> 2, ok && (n < lim));
      U4 b = B::g_r128p(base, (e0 >> 2) + 1, ok && ((n + 4) < lim));
      v[0] = B::as_f32(a.x); v[1] = B::as_f32(a.y); v[2] = B::as_f32(a.z); v[3] = B::as_f32(a.w);
      v[4] = B::as_f32(b.x); v[5] = B::as_f32(b.y); v[6] = B::as_f32(b.z); v[7] = B::as_f32(b.w);
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = B::as_f32(B::g_r32(base, e0 + q, ok && ((n + q) < lim)));
    }
  }
  // 16-byte path of k_rows_in: every load of the wave's slice first (clamped, unconditional), then the rounding and the LDS writes.
  // The loop below it consumes each chunk's two loads before it requests the next chunk's: 8 memory round trips in a row for the
  // filter row of a head -- in the forward launch that transforms its own filter (kfft_head) that was most of the step's head time.
  static FFC_FN void k_rows_in_fast(const KfArgs& a, int unit_id, Unit un) {
    const i32 lane = B::opaque(B::lane());
    U4 A[BD::NCH], Bq[BD::NCH];
#pragma unroll
    for (int i = 0; i < BD::NCH; i++) {
      // (no skip for chunks beyond Lk: a branch per chunk would put every chunk's loads into a flow block of their own, with a
      // conservative s_waitcnt at each merge; the clamped index makes such a chunk re-read one 16-byte piece)
      i32 idx = lane + i * 64;
      i32 row = idx / BD::CPR, m = (idx % BD::CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
      i32 hd, n;
      if constexpr (GEO::OUTER) { hd = row * 0 + unit_id; n = row * GEO::Mi + m; }
      else { hd = row + unit_id * GEO::G; n = m; }
      i32 hb = B::imin(hd, a.H - 1) * a.Lk;
      A[i] = B::g_r128(a.k, (hb + B::imin(n, a.Lk - 4)) >> 2);
      Bq[i] = B::g_r128(a.k, (hb + B::imin(n + 4, a.Lk - 4)) >> 2);
    }
    B::sched_fence();
#pragma unroll
    for (int i = 0; i < BD::NCH; i++) {
      i32 idx = lane + i * 64;
      i32 row = idx / BD::CPR, m = (idx % BD::CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
      pred sw;
      i32 off = BD::pair_off(row, m, &sw) + un.eb;
      i32 hd, n;
      if constexpr (GEO::OUTER) { hd = row * 0 + unit_id; n = row * GEO::Mi + m; }
      else { hd = row + unit_id * GEO::G; n = m; }
      U4 o;
      if (GEO::OUTER && ((i * 64) / BD::CPR) * GEO::Mi >= a.Lk) {
        o.x = B::uconst(0); o.y = B::uconst(0); o.z = B::uconst(0); o.w = B::uconst(0);
      } else {
        pred oka = (hd < a.H) && (n < a.Lk), okb = (hd < a.H) && ((n + 4) < a.Lk);
        const u32 zz = B::uconst(0);
        f32 v[8];
        v[0] = B::as_f32(B::sel(oka, A[i].x, zz)); v[1] = B::as_f32(B::sel(oka, A[i].y, zz));
        v[2] = B::as_f32(B::sel(oka, A[i].z, zz)); v[3] = B::as_f32(B::sel(oka, A[i].w, zz));
        v[4] = B::as_f32(B::sel(okb, Bq[i].x, zz)); v[5] = B::as_f32(B::sel(okb, Bq[i].y, zz));
        v[6] = B::as_f32(B::sel(okb, Bq[i].z, zz)); v[7] = B::as_f32(B::sel(okb, Bq[i].w, zz));
#pragma unroll
        for (int q8 = 0; q8 < 8; q8++) v[q8] = v[q8] * a.prescale;
        u32 p0 = B::template pack<DT>(v[0], v[1]), p1 = B::template pack<DT>(v[2], v[3]);
        u32 p2 = B::template pack<DT>(v[4], v[5]), p3 = B::template pack<DT>(v[6], v[7]);
        o.x = B::sel(sw, p2, p0); o.y = B::sel(sw, p3, p1); o.z = B::sel(sw, p0, p2); o.w = B::sel(sw, p1, p3);
      }
      B::lds_w128(off, o, B::ptrue());
      U4 z; z.x = B::uconst(0); z.y = B::uconst(0); z.z = B::uconst(0); z.w = B::uconst(0);
      B::lds_w128(off + GEO::PLANE, z, B::ptrue());
    }
  }
  static FFC_FN void k_rows_in(const KfArgs& a, int unit_id, Unit un) {
    if (a.fast) { k_rows_in_fast(a, unit_id, un); return; }
    const i32 lane = B::opaque(B::lane());
#pragma unroll
    for (int i = 0; i < BD::NCH; i++) {
      i32 idx = lane + i * 64;
      i32 row = idx / BD::CPR, m = (idx % BD::CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
      pred sw;
      i32 off = BD::pair_off(row, m, &sw) + un.eb;
      i32 hd, n;
      if constexpr (GEO::OUTER) { hd = row * 0 + unit_id; n = row * GEO::Mi + m; }
      else { hd = row + unit_id * GEO::G; n = m; }
      pred ok = hd < a.H;
      f32 v[8];
      fload8(a.k, hd * a.Lk + n, n, a.Lk, a.fast != 0, ok, v);
#pragma unroll
      for (int q8 = 0; q8 < 8; q8++) v[q8] = v[q8] * a.prescale;
      U4 o;
      u32 p0 = B::template pack<DT>(v[0], v[1]), p1 = B::template pack<DT>(v[2], v[3]);
      u32 p2 = B::template pack<DT>(v[4], v[5]), p3 = B::template pack<DT>(v[6], v[7]);
      o.x = B::sel(sw, p2, p0); o.y = B::sel(sw, p3, p1); o.z = B::sel(sw, p0, p2); o.w = B::sel(sw, p1, p3);
      B::lds_w128(off, o, B::ptrue());
      U4 z; z.x = B::uconst(0); z.y = B::uconst(0); z.z = B::uconst(0); z.w = B::uconst(0);
      B::lds_w128(off + GEO::PLANE, z, B::ptrue());
    }
  }
  // multi-pass sizes: E row n1 = sum_n0 W_R^{n0 k0} k[n0 M + n1 Mi + m] (fp32 sums; (-i)^q (w + 0i) = (w,0),(0,-w),(-w,0),(0,w))
  static FFC_FN void k_rows_in_rp(const KfArgs& a, int unit_id, Unit un, Pass ps) {
    const i32 lane = B::opaque(B::lane());
    const int n0max = (a.Lk + GEO::N - 1) / GEO::N;
#pragma unroll
    for (int i = 0; i < BD::NCH; i++) {
      i32 idx = lane + i * 64;
      i32 row = idx / BD::CPR, m = (idx % BD::CPR) * 8 + un.wq * 128 * GEO::S1;
      pred sw;
      i32 off = BD::pair_off(row, m, &sw) + un.eb;
      i32 hd = row * 0 + unit_id, n = row * GEO::Mi + m;
      pred ok = hd < a.H;
      f32 vr[8], vi[8];
#pragma unroll
      for (int q8 = 0; q8 < 8; q8++) { vr[q8] = B::fconst(0.f); vi[q8] = B::fconst(0.f); }
#pragma unroll 1
      for (int n0 = 0; n0 < n0max; n0++) {
        f32 w[8];
        fload8(a.k, hd * a.Lk + n + n0 * GEO::N, n + n0 * GEO::N, a.Lk, a.fast != 0, ok, w);
        const int q = (n0 * ps.k0 * (4 / ps.R)) & 3;
        const float s = (q == 0 || q == 3) ? a.prescale : -a.prescale;
#pragma unroll
        for (int q8 = 0; q8 < 8; q8++) {
          if (q & 1) vi[q8] = vi[q8] + w[q8] * s;
          else vr[q8] = vr[q8] + w[q8] * s;
        }
      }
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        const f32* v = pl ? vi : vr;
        u32 p0 = B::template pack<DT>(v[0], v[1]), p1 = B::template pack<DT>(v[2], v[3]);
        u32 p2 = B::template pack<DT>(v[4], v[5]), p3 = B::template pack<DT>(v[6], v[7]);
        U4 o;
        o.x = B::sel(sw, p2, p0); o.y = B::sel(sw, p3, p1); o.z = B::sel(sw, p0, p2); o.w = B::sel(sw, p1, p3);
        B::lds_w128(off + pl * GEO::PLANE, o, B::ptrue());
      }
    }
  }
  static FFC_FN void kf_store(const KfArgs& a, int unit_id, int tau, const A16& re, const A16& im, int hmul = 1) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      U4 v;
      u32 w[4];
#pragma unroll
      for (int q = 0; q < 4; q++) w[q] = B::template pack<DT>(re[4 * rq + q] * a.scale, im[4 * rq + q] * a.scale);
      v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
      if constexpr (GEO::OUTER) {
        pred ok = (c * 0 + unit_id) < a.H * hmul;      // hmul: rows are (head, pass) for the multi-pass sizes
        i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c) + unit_id * (GEO::NT * 256);
        B::g_w128(a.kf, idx, v, ok);
      } else {
        // the tile holds G heads (sub-blocks); each head's k_f tile replicates its block G times
        i32 V = hi * 4 + 8 * rq;
        i32 sV = V / GEO::N3, k3 = V % GEO::N3, sU = c / GEO::N2, k2 = c % GEO::N2;
        i32 hd = sU * GEO::SV + sV + unit_id * GEO::G;
        pred ok = hd < a.H;
#pragma unroll
        for (int su = 0; su < GEO::SU; su++)
#pragma unroll
          for (int sv = 0; sv < GEO::SV; sv++) {
            i32 rho = (k3 + sv * GEO::N3) >> 2;
            i32 idx = (rho * 32 + (k2 + su * GEO::N2)) + hd * 256;
            B::g_w128(a.kf, idx, v, ok);
          }
      }
    }
  }
  // one 1024-point tile of k_f row `hrow` (G == 1 geometries: the tile is the row; same layout as the OUTER branch of kf_store)
  static FFC_FN void kf_store_flat(const KfArgs& a, int hrow, const A16& re, const A16& im, int hlim) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      U4 v;
      v.x = B::template pack<DT>(re[4 * rq] * a.scale, im[4 * rq] * a.scale);
      v.y = B::template pack<DT>(re[4 * rq + 1] * a.scale, im[4 * rq + 1] * a.scale);
      v.z = B::template pack<DT>(re[4 * rq + 2] * a.scale, im[4 * rq + 2] * a.scale);
      v.w = B::template pack<DT>(re[4 * rq + 3] * a.scale, im[4 * rq + 3] * a.scale);
      B::g_w128(a.kf, ((hi + 2 * rq) * 32 + c) + hrow * 256, v, (c * 0 + hrow) < hlim);
    }
  }
  // one workgroup: UPW units (heads, or tiles of G heads)
  static FFC_FN void kfft(const KfArgs& a, int wg) {
    BD::setup_tables(a.tab, a.t);
    const int wv = B::wave();
    Unit un;
    un.wq = wv % GEO::NW;
    const int u = wv / GEO::NW;
    un.eb = u * GEO::EBYTES;
    const int unit_id = wg * GEO::UPW + u;
    const int nunits = GEO::OUTER ? a.H : (a.H + GEO::G - 1) / GEO::G;
    const bool act = unit_id < nunits;
    InnerRegs R;
    BD::load_inner(R, un);
    if constexpr (GEO::N == 32768) {
      if (a.R > 1) {       // multi-pass size: k_f rows (head, k0) = the spectrum samples f = k0 (mod R)
#pragma unroll 1
        for (int k0 = 0; k0 < a.R; k0++) {
          const Pass ps = make_pass(a.tab, a.t, a.R, k0);
          if (act) {
            if (a.xpair) {       // complex input (pair-plane tensor (2, H, R*M)): inner k_f rows of the big FFT sizes
              ConvArgs cv{};
              cv.u = a.xpair; cv.B = 2; cv.H = a.H; cv.L = GEO::N * a.R; cv.fast = 1; cv.sbu = (int64_t)a.H * cv.L;
              BD::template rows_in_rp<BD::NCH>(cv, unit_id, 0, un, ps);
            } else {
              k_rows_in_rp(a, unit_id, un, ps);
            }
            B::lds_fence();
            BD::template outer_stage<true, false, true>(a.Lk, un, a.s_fwd, ps);
          }
          B::barrier();
          if (act) {
#pragma unroll 1
            for (int tt = 0; tt < GEO::TPW; tt++) {
              A16 re, im;
              BD::tile_fwd(un.wq * GEO::TPW + tt, R, un, re, im);
              kf_store(a, unit_id * a.R + k0, un.wq * GEO::TPW + tt, re, im, a.R);
            }
          }
          B::barrier();
        }
        return;
      }
    }
    if constexpr (GEO::OUTER) {
      if (act) {
        if (a.xpair) {
          ConvArgs cv{};
          cv.u = a.xpair; cv.B = 2; cv.H = a.H; cv.L = GEO::N; cv.fast = 1; cv.sbu = (int64_t)a.H * GEO::N;
          BD::rows_in(cv, unit_id, 0, un);
        } else {
          k_rows_in(a, unit_id, un);
        }
        B::lds_fence();
        if ((GEO::N1 / 2) * GEO::Mi >= a.Lk) BD::template outer_stage<true, true>(a.Lk, un, a.s_fwd);
        else BD::template outer_stage<true, false>(a.Lk, un, a.s_fwd);
      }
      B::barrier();
      if (act) {
#pragma unroll 1
        for (int tt = 0; tt < GEO::TPW; tt++) {
          A16 re, im;
          BD::tile_fwd(un.wq * GEO::TPW + tt, R, un, re, im);
          kf_store(a, unit_id, un.wq * GEO::TPW + tt, re, im);
        }
      }
    } else if (a.R > 1) {
      // inner-only multi-pass form (fft 2048 on the 32 x 32 kernel, BD::InnerPass): k_f rows (head, k0)
      if constexpr (GEO::N == 1024) {
        BD::setup_tables_ipass(a.tab, a.t, a.R);
        if (act) {
#pragma unroll 1
          for (int k0 = 0; k0 < a.R; k0++) {
            Pass ps; ps.k0 = k0; ps.R = a.R;
            InnerPass ip;
            BD::load_inner_pass(ip, k0);
            k_rows_in_rp(a, unit_id, un, ps);
            B::lds_fence();
            A16 re, im;
            BD::template tile_fwd<true, true, false, FFC_IP_LEAN != 0>(0, R, un, re, im, &ip);
            kf_store_flat(a, unit_id * a.R + k0, re, im, a.H * a.R);
            B::lds_fence();
          }
        }
      }
    } else {
      if (act) {
        k_rows_in(a, unit_id, un);
        B::lds_fence();
        A16 re, im;
        BD::tile_fwd(0, R, un, re, im);
        kf_store(a, unit_id, 0, re, im);
      }
    }
  }

  // k -> k_f of ONE head inside another kernel's workgroup (ConvArgs::kfuse_k: the forward kernel when the workgroup owns its
  // head): the single-pass OUTER branch of kfft() for unit `h`.  The plan tables are in LDS already.  Ends with a barrier: the
  // caller's row copies may overwrite the exchange buffer.
  static FFC_FN void kfft_head(const ConvArgs& c, int h) {
    static_assert(GEO::OUTER && GEO::NW > 1, "kfft_head: sizes whose waves meet at workgroup barriers anyway");
    // (several units per workgroup, fft 8192 / 16384: unit 0's waves transform the filter in their exchange buffer, the others
    // wait at the two barriers; every wave of the workgroup later reads the tiles from global memory -- __syncthreads orders them)
    const int unit = B::wave() / GEO::NW;
    KfArgs a{};
    a.k = c.kfuse_k; a.kf = const_cast<void*>(c.kf); a.H = c.H; a.Lk = c.kfuse_Lk; a.scale = c.kfuse_scale; a.s_fwd = c.s_fwd;
    a.prescale = DT == DT_F16 ? 256.f : 1.f;      // as ffc_kernel_fft: fp16 plans scale k up before rounding it (kfuse_scale carries 1 / 256)
    a.fast = c.kfuse_fast;
    Unit un;
    un.wq = B::wave() % GEO::NW;
    un.eb = 0;
    if (unit == 0) {
      if (c.kfuse_x) {         // complex rows (pair-plane tensor): the k -> k_f step of the HBM-level sizes (kfft()'s xpair branch)
        ConvArgs cv{};
        cv.u = c.kfuse_x; cv.B = 2; cv.H = c.H; cv.L = GEO::N; cv.fast = 1; cv.sbu = (int64_t)c.H * GEO::N;
        BD::rows_in(cv, h, 0, un);
        a.Lk = GEO::N; a.prescale = 1.0f;
      } else {
        k_rows_in(a, h, un);
      }
      B::lds_fence();
      if ((GEO::N1 / 2) * GEO::Mi >= a.Lk) BD::template outer_stage<true, true>(a.Lk, un, a.s_fwd);
      else BD::template outer_stage<true, false>(a.Lk, un, a.s_fwd);
    }
    B::barrier();
    if (unit == 0) {
      InnerRegs R;
      BD::load_inner(R, un);
#pragma unroll 1
      for (int tt = 0; tt < GEO::TPW; tt++) {
        A16 re, im;
        BD::tile_fwd(un.wq * GEO::TPW + tt, R, un, re, im);
        kf_store(a, h, un.wq * GEO::TPW + tt, re, im);
      }
    }
    B::barrier();
  }

  // ------------------------------------------------------------------ dk_f accumulation
  struct ZReg { u32 r[8], i[8]; };   // a spectrum tile as packed dtype pairs (plain dwords: no register-tuple constraint)
  static FFC_FN void z_pack(const A16& re, const A16& im, ZReg& z) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      z.r[q] = B::template pack<DT>(re[2 * q], re[2 * q + 1]);
      z.i[q] = B::template pack<DT>(im[2 * q], im[2 * q + 1]);
    }
  }
  using BD::z_store;
  using BD::z_load;
  // previous partial sums of a tile, issued at the start of the tile so the latency overlaps tile_fwd
  struct WOld { U4 v[4][2]; };
  static FFC_FN void w_load_old(const float* slab, int tau, bool first, WOld& o) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c) * 2;
      o.v[rq][0] = B::g_r128p(slab, idx, !first ? B::ptrue() : B::pfalse());
      o.v[rq][1] = B::g_r128p(slab, idx + 1, !first ? B::ptrue() : B::pfalse());
    }
  }
  static FFC_FN void w_update(float* slab, int tau, const WOld& o, const typename BD::KfRegs& zv, const A16& re, const A16& im) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c) * 2;
      u32 wv[4] = {zv.v[rq].x, zv.v[rq].y, zv.v[rq].z, zv.v[rq].w};
      f32 wr[4], wi[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 4 * rq + q;
        f32 ur = B::template unpack_lo<DT>(wv[q]), ui = B::template unpack_hi<DT>(wv[q]);
        wr[q] = re[r] * ur + im[r] * ui;
        wi[q] = im[r] * ur - re[r] * ui;
      }
      const U4& o0 = o.v[rq][0]; const U4& o1 = o.v[rq][1];       // zeros on the first pair
      wr[0] = wr[0] + B::as_f32(o0.x); wi[0] = wi[0] + B::as_f32(o0.y);
      wr[1] = wr[1] + B::as_f32(o0.z); wi[1] = wi[1] + B::as_f32(o0.w);
      wr[2] = wr[2] + B::as_f32(o1.x); wi[2] = wi[2] + B::as_f32(o1.y);
      wr[3] = wr[3] + B::as_f32(o1.z); wi[3] = wi[3] + B::as_f32(o1.w);
      U4 n0, n1;
      n0.x = B::as_u32(wr[0]); n0.y = B::as_u32(wi[0]); n0.z = B::as_u32(wr[1]); n0.w = B::as_u32(wi[1]);
      n1.x = B::as_u32(wr[2]); n1.y = B::as_u32(wi[2]); n1.z = B::as_u32(wr[3]); n1.w = B::as_u32(wi[3]);
      B::g_w128(slab, idx, n0, B::ptrue());
      B::g_w128(slab, idx + 1, n1, B::ptrue());
    }
  }
  // W (+)= Zd * conj(Zv), Zv given as (re,im)-interleaved dtype pairs (KfRegs layout)
  static FFC_FN void w_accum_z(float* slab, int tau, bool first, const typename BD::KfRegs& zv, const A16& re, const A16& im) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c) * 2;
      u32 wv[4] = {zv.v[rq].x, zv.v[rq].y, zv.v[rq].z, zv.v[rq].w};
      f32 wr[4], wi[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 4 * rq + q;
        f32 ur = B::template unpack_lo<DT>(wv[q]), ui = B::template unpack_hi<DT>(wv[q]);
        wr[q] = re[r] * ur + im[r] * ui;
        wi[q] = im[r] * ur - re[r] * ui;
      }
      if (!first) {
        U4 o0 = B::g_r128(slab, idx), o1 = B::g_r128(slab, idx + 1);
        wr[0] = wr[0] + B::as_f32(o0.x); wi[0] = wi[0] + B::as_f32(o0.y);
        wr[1] = wr[1] + B::as_f32(o0.z); wi[1] = wi[1] + B::as_f32(o0.w);
        wr[2] = wr[2] + B::as_f32(o1.x); wi[2] = wi[2] + B::as_f32(o1.y);
        wr[3] = wr[3] + B::as_f32(o1.z); wi[3] = wi[3] + B::as_f32(o1.w);
      }
      U4 n0, n1;
      n0.x = B::as_u32(wr[0]); n0.y = B::as_u32(wi[0]); n0.z = B::as_u32(wr[1]); n0.w = B::as_u32(wi[1]);
      n1.x = B::as_u32(wr[2]); n1.y = B::as_u32(wi[2]); n1.z = B::as_u32(wr[3]); n1.w = B::as_u32(wi[3]);
      B::g_w128(slab, idx, n0, B::ptrue());
      B::g_w128(slab, idx + 1, n1, B::ptrue());
    }
  }
  static FFC_FN void w_accum(float* slab, int tau, bool first, const ZReg& zv, const A16& re, const A16& im) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c) * 2;   // 16-byte units: 2 complex fp32 each
      f32 wr[4], wi[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 4 * rq + q;
        u32 pr = zv.r[r >> 1], pi = zv.i[r >> 1];
        f32 ur = (r & 1) ? B::template unpack_hi<DT>(pr) : B::template unpack_lo<DT>(pr);
        f32 ui = (r & 1) ? B::template unpack_hi<DT>(pi) : B::template unpack_lo<DT>(pi);
        // Zd * conj(Zv)
        wr[q] = re[r] * ur + im[r] * ui;
        wi[q] = im[r] * ur - re[r] * ui;
      }
      if (!first) {
        U4 o0 = B::g_r128(slab, idx), o1 = B::g_r128(slab, idx + 1);
        wr[0] = wr[0] + B::as_f32(o0.x); wi[0] = wi[0] + B::as_f32(o0.y);
        wr[1] = wr[1] + B::as_f32(o0.z); wi[1] = wi[1] + B::as_f32(o0.w);
        wr[2] = wr[2] + B::as_f32(o1.x); wi[2] = wi[2] + B::as_f32(o1.y);
        wr[3] = wr[3] + B::as_f32(o1.z); wi[3] = wi[3] + B::as_f32(o1.w);
      }
      U4 n0, n1;
      n0.x = B::as_u32(wr[0]); n0.y = B::as_u32(wi[0]); n0.z = B::as_u32(wr[1]); n0.w = B::as_u32(wi[1]);
      n1.x = B::as_u32(wr[2]); n1.y = B::as_u32(wi[2]); n1.z = B::as_u32(wr[3]); n1.w = B::as_u32(wi[3]);
      B::g_w128(slab, idx, n0, B::ptrue());
      B::g_w128(slab, idx + 1, n1, B::ptrue());
    }
  }
  static FFC_FN void w_zero(float* slab, int tau) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
    U4 z; z.x = B::uconst(0); z.y = B::uconst(0); z.z = B::uconst(0); z.w = B::uconst(0);
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c) * 2;
      B::g_w128(slab, idx, z, B::ptrue());
      B::g_w128(slab, idx + 1, z, B::ptrue());
    }
  }
  // Register-resident partial sums.  A wave owns its TPW tiles of W for the whole chunk: 4 tiles x 32 fp32
  // per lane = 128 registers, which live in the accumulation half (AGPRs a0..a127) of the unified register
  // file, addressed explicitly by the backend (agpr_get / agpr_set).  They are invisible to the register
  // allocator, which keeps the architectural half (128 VGPRs) for the transforms; the slab is written once
  // per chunk instead of read-modified-written for every pair.  FFC_WREG=0 selects the slab path.
  using F2 = typename B::F2;       // fp32 pair = one packed-math register pair
  static constexpr int WREG = FFC_WREG < GEO::TPW ? 0 : GEO::TPW;
  struct WAcc { int unused; };
  template <int I0, int N>
  static FFC_FN void w_acc_zero_range() {
    if constexpr (N == 1) B::template agpr_set<I0>(B::fconst(0.f));
    else { w_acc_zero_range<I0, N / 2>(); w_acc_zero_range<I0 + N / 2, N - N / 2>(); }
  }
  static FFC_FN void w_acc_zero(WAcc&) {
    if constexpr (WREG > 0) { B::agpr_reserve(); w_acc_zero_range<0, 32 * WREG>(); }
  }
  // accumulator a[32T + 16*part + r]: part 0 = re, 1 = im, r = accumulator row slot
  template <int T, int RQ>
  static FFC_FN void w_acc_quarter(const U4& z, const A16& re, const A16& im) {
    u32 wv[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
    for (int j = 0; j < 2; j++) {
      F2 zr = B::f2(B::template unpack_lo<DT>(wv[2 * j]), B::template unpack_lo<DT>(wv[2 * j + 1]));
      F2 zi = B::f2(B::template unpack_hi<DT>(wv[2 * j]), B::template unpack_hi<DT>(wv[2 * j + 1]));
      if (j == 0) {
        constexpr int R0 = 4 * RQ;
        F2 wr = B::f2(B::template agpr_get<32 * T + R0>(), B::template agpr_get<32 * T + R0 + 1>());
        F2 wi = B::f2(B::template agpr_get<32 * T + 16 + R0>(), B::template agpr_get<32 * T + 16 + R0 + 1>());
        B::cmac2_conj(wr, wi, re, im, R0, zr, zi);
        B::template agpr_set<32 * T + R0>(B::f2_lo(wr)); B::template agpr_set<32 * T + R0 + 1>(B::f2_hi(wr));
        B::template agpr_set<32 * T + 16 + R0>(B::f2_lo(wi)); B::template agpr_set<32 * T + 16 + R0 + 1>(B::f2_hi(wi));
      } else {
        constexpr int R0 = 4 * RQ + 2;
        F2 wr = B::f2(B::template agpr_get<32 * T + R0>(), B::template agpr_get<32 * T + R0 + 1>());
        F2 wi = B::f2(B::template agpr_get<32 * T + 16 + R0>(), B::template agpr_get<32 * T + 16 + R0 + 1>());
        B::cmac2_conj(wr, wi, re, im, R0, zr, zi);
        B::template agpr_set<32 * T + R0>(B::f2_lo(wr)); B::template agpr_set<32 * T + R0 + 1>(B::f2_hi(wr));
        B::template agpr_set<32 * T + 16 + R0>(B::f2_lo(wi)); B::template agpr_set<32 * T + 16 + R0 + 1>(B::f2_hi(wi));
      }
    }
  }
  // Round 6 experiment (FFC_WACC_MFMA = 1; measured, NOT adopted: profiles/r06_ab_wacc_mfma.txt): the sums accumulated by the MATRIX pipe.
  // The fp32 products P = D (x) conj Z of a tile are rounded to bf16 MFMA operands (to_op form: operand dword d of K-step ms =
  // pack(P[8 ms + 2 d], P[8 ms + 2 d + 1]), whose contraction slot (ms, lane half, e) carries accumulator row kslot_row(ms, hi, e),
  // ffc_layout.h) and multiplied by a permuted identity straight into a[32 T ..]: W += I x P, exact fp32 accumulation of bf16-rounded
  // products.  It removes the 128 v_accvgpr_read / _write + 16 packed adds per tile and pair of the VALU form for 16 v_cvt_pk + ~25 VALU
  // (identity operands) + 4 MFMAs -- about 350 issue cycles per tile on the r03 cost table -- and the backward kernels run EXACTLY as
  // fast as before (config 2: 0.6222 / 0.6236 / 0.6207 against 0.6199 / 0.6081 / 0.6236 ms, same box, interleaved; fft 4096 5 % slower):
  // the accumulation is not on the critical path of the pair loop.  Parity-green on the simulator (209 cases) and the GPU.  Default 0.
#ifndef FFC_WACC_MFMA
#define FFC_WACC_MFMA 0
#endif
  // A operand of the identity K-step ms: lane (i = lane & 31, hi' = lane >> 5), slot e: 1.0 iff i == kslot_row(ms, hi', e)
  static FFC_FN W4 ident_op(int ms) {
    const i32 lane = B::opaque(B::lane());
    const i32 t = (lane & 31) - (lane >> 5) * 4 - 16 * ms;      // = 8 (e >> 2) + (e & 3) for the matching slot, if any
    const pred ok = (t >= 0) && ((t & 4) < 1) && (t < 12);
    const i32 e = (t & 3) + ((t >> 3) << 2);
    const u32 one = B::sel((e & 1) >= 1, B::uconst(0x3F800000u), B::uconst(0x00003F80u));
    W4 w;
#pragma unroll
    for (int d = 0; d < 4; d++) w[d] = B::sel(ok && ((e >> 1) < d + 1) && ((e >> 1) >= d), one, B::uconst(0));
    return w;
  }
  // products of accumulator rows 4 RQ .. 4 RQ + 3 with the conjugated spectrum quad z -> two operand dwords per component
  static FFC_FN void w_prod_quarter(int RQ, const U4& z, const A16& re, const A16& im, u32 (&pr)[2], u32 (&pi)[2]) {
    u32 wv[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int r0 = 4 * RQ + 2 * j;
      f32 zr0 = B::template unpack_lo<DT>(wv[2 * j]), zi0 = B::template unpack_hi<DT>(wv[2 * j]);
      f32 zr1 = B::template unpack_lo<DT>(wv[2 * j + 1]), zi1 = B::template unpack_hi<DT>(wv[2 * j + 1]);
      f32 p0 = re[r0] * zr0 + im[r0] * zi0, p1 = re[r0 + 1] * zr1 + im[r0 + 1] * zi1;
      f32 q0 = im[r0] * zr0 - re[r0] * zi0, q1 = im[r0 + 1] * zr1 - re[r0 + 1] * zi1;
      pr[j] = B::template pack<DT_BF16>(p0, p1);
      pi[j] = B::template pack<DT_BF16>(q0, q1);
    }
  }
  template <int T>
  static FFC_FN void w_acc_tile(const typename BD::KfRegs& zv, const A16& re, const A16& im) {
#if defined(FFC_KO) && (FFC_KO & 128)
    return;        // knock-out timing experiment: no dk_f accumulation
#endif
    if constexpr (FFC_WACC_MFMA != 0) {
#pragma unroll
      for (int ms = 0; ms < 2; ms++) {
        u32 r01[2], i01[2], r23[2], i23[2];
        w_prod_quarter(2 * ms, zv.v[2 * ms], re, im, r01, i01);
        w_prod_quarter(2 * ms + 1, zv.v[2 * ms + 1], re, im, r23, i23);
        const W4 id = ident_op(ms);
        B::template mfma_acc_bf16<32 * T>(id, B::w4(r01[0], r01[1], r23[0], r23[1]));
        B::template mfma_acc_bf16<32 * T + 16>(id, B::w4(i01[0], i01[1], i23[0], i23[1]));
      }
      return;
    }
    w_acc_quarter<T, 0>(zv.v[0], re, im);
    w_acc_quarter<T, 1>(zv.v[1], re, im);
    w_acc_quarter<T, 2>(zv.v[2], re, im);
    w_acc_quarter<T, 3>(zv.v[3], re, im);
  }
  template <int T, int RQ>
  static FFC_FN void w_acc_store_q(float* slab, int tau0, i32 hi, i32 c) {
    i32 idx = ((hi + ((tau0 + T) * 8 + 2 * RQ)) * 32 + c) * 2;
    constexpr int A0 = 32 * T + 4 * RQ;
    U4 n0, n1;
    n0.x = B::as_u32(B::template agpr_get<A0>());     n0.y = B::as_u32(B::template agpr_get<A0 + 16>());
    n0.z = B::as_u32(B::template agpr_get<A0 + 1>()); n0.w = B::as_u32(B::template agpr_get<A0 + 17>());
    n1.x = B::as_u32(B::template agpr_get<A0 + 2>()); n1.y = B::as_u32(B::template agpr_get<A0 + 18>());
    n1.z = B::as_u32(B::template agpr_get<A0 + 3>()); n1.w = B::as_u32(B::template agpr_get<A0 + 19>());
    B::g_w128(slab, idx, n0, B::ptrue());
    B::g_w128(slab, idx + 1, n1, B::ptrue());
  }
  template <int T>
  static FFC_FN void w_acc_store_t(float* slab, int tau0, i32 hi, i32 c) {
    w_acc_store_q<T, 0>(slab, tau0, hi, c); w_acc_store_q<T, 1>(slab, tau0, hi, c);
    w_acc_store_q<T, 2>(slab, tau0, hi, c); w_acc_store_q<T, 3>(slab, tau0, hi, c);
  }
  static FFC_FN void w_acc_store(float* slab, int tau0, const WAcc&) {
    if constexpr (WREG > 0) {
      const i32 lane = B::opaque(B::lane());
      const i32 c = lane & 31, hi = lane >> 5;
      w_acc_store_t<0>(slab, tau0, hi, c); w_acc_store_t<1>(slab, tau0, hi, c);
      w_acc_store_t<2>(slab, tau0, hi, c); w_acc_store_t<3>(slab, tau0, hi, c);
    }
  }
  // Cross-unit reduction of the accumulation registers (UPW > 1: the UPW units of a workgroup are UPW pairs of the
  // SAME head, each with its own W in a0..a127).  Per tile: every wave parks its 32 accumulators in the (now idle)
  // exchange buffers, [wave][register][lane] fp32; after a barrier the UPW waves that share a tile each sum one
  // 1/UPW share over the units in a fixed order (bitwise reproducible) and write it to the chunk's single slab.
  // One slab per chunk instead of UPW: 8x less dk_f traffic at fft 4096 (written here, read back by dkifft).
  template <int T, int R0, int NR>
  static FFC_FN void w_acc_park(i32 base) {
    if constexpr (NR == 1) B::lds_w32(base + R0 * 256, B::as_u32(B::template agpr_get<32 * T + R0>()));
    else { w_acc_park<T, R0, NR / 2>(base); w_acc_park<T, R0 + NR / 2, NR - NR / 2>(base); }
  }
  template <int T>
  static FFC_FN void w_acc_reduce_tile(float* slab, int tau0, int u, int wq, i32 lane) {
    const i32 c = lane & 31, hi = lane >> 5;
    w_acc_park<T, 0, 32>(GEO::L_E + (u * GEO::NW + wq) * 8192 + lane * 4);
    B::barrier();
    constexpr int SPW = 8 / GEO::UPW;          // (RQ, j) slots of 2 complex values per wave
#pragma unroll
    for (int qq = 0; qq < SPW; qq++) {
      const int q = u * SPW + qq;
      const int RQ = q >> 1, j = q & 1;
      const int r0 = 4 * RQ + 2 * j;
      f32 sre0 = B::fconst(0.f), sre1 = B::fconst(0.f), sim0 = B::fconst(0.f), sim1 = B::fconst(0.f);
#pragma unroll
      for (int s2 = 0; s2 < GEO::UPW; s2++) {
        const i32 src = GEO::L_E + (s2 * GEO::NW + wq) * 8192 + lane * 4;
        sre0 = sre0 + B::as_f32(B::lds_r32(src + r0 * 256));
        sre1 = sre1 + B::as_f32(B::lds_r32(src + (r0 + 1) * 256));
        sim0 = sim0 + B::as_f32(B::lds_r32(src + (16 + r0) * 256));
        sim1 = sim1 + B::as_f32(B::lds_r32(src + (17 + r0) * 256));
      }
      U4 n;
      n.x = B::as_u32(sre0); n.y = B::as_u32(sim0); n.z = B::as_u32(sre1); n.w = B::as_u32(sim1);
      i32 idx = ((hi + ((tau0 + T) * 8 + 2 * RQ)) * 32 + c) * 2 + j;
      B::g_w128(slab, idx, n, B::ptrue());
    }
    B::barrier();
  }
  // end of a chunk (OUTER geometries): `slab` is the chunk's single slab of this head, [chunk][H][NT*2048] floats
  static FFC_FN void w_acc_finish(float* slab, int u, Unit un, WAcc& W) {
    if constexpr (GEO::UPW == 1) {
      w_acc_store(slab, un.wq * GEO::TPW, W);
    } else {
      static_assert(WREG == GEO::TPW, "the per-pair slab path (FFC_WREG=0) only exists for one unit per workgroup");
      static_assert(GEO::UPW * GEO::NW * 8192 <= GEO::UPW * GEO::EBYTES, "parking area fits the exchange buffers");
      const i32 lane = B::opaque(B::lane());
      B::barrier();
      w_acc_reduce_tile<0>(slab, un.wq * GEO::TPW, u, un.wq, lane);
      w_acc_reduce_tile<1>(slab, un.wq * GEO::TPW, u, un.wq, lane);
      w_acc_reduce_tile<2>(slab, un.wq * GEO::TPW, u, un.wq, lane);
      w_acc_reduce_tile<3>(slab, un.wq * GEO::TPW, u, un.wq, lane);
    }
  }
  // ---- dk from the accumulation registers (DkfArgs::dk_out): the tail of the fused backward kernel when the workgroup owns the head
  template <int T, int R0, int NR>
  static FFC_FN void w_acc_read(float sc, A16& re, A16& im) {
    if constexpr (NR == 1) {
      re[R0] = B::template agpr_get<32 * T + R0>() * sc;
      im[R0] = B::template agpr_get<32 * T + 16 + R0>() * sc;
    } else {
      w_acc_read<T, R0, NR / 2>(sc, re, im);
      w_acc_read<T, R0 + NR / 2, NR - NR / 2>(sc, re, im);
    }
  }
  static FFC_FN Unit unit_of(int u, int wq) { Unit un; un.wq = wq; un.eb = u * GEO::EBYTES; return un; }
  // rows of the inverted dk_f: real part -> dk (H, Lk) fp32, or both planes -> the pair-plane tensor (DkfArgs::dk_pair)
  static FFC_FN void dk_tail_rows(const DkfArgs& d, int h, Unit un) {
    if (d.dk_pair) {
      ConvArgs cv{};
      cv.y = d.dk_pair; cv.B = 2; cv.H = d.c.H; cv.L = GEO::N; cv.fast = 1; cv.sby = (int64_t)d.c.H * GEO::N;
      BD::rows_out(cv, h, 0, un);
    } else {
      DkArgs ka{};
      ka.dk = d.dk_out; ka.H = d.c.H; ka.Lk = d.Lk; ka.fast = d.dk_fast;
      dk_rows_out(ka, h, un);
    }
  }
  template <int T>
  static FFC_FN void dk_tail_tile(const DkfArgs& d, Unit un, const InnerRegs& R) {
    A16 re, im;
    w_acc_read<T, 0, 16>(d.dk_scale, re, im);
    BD::template tile_inv<false>(d.c.s_inv, un.wq * GEO::TPW + T, R, un, re, im);
  }
  static FFC_FN void dk_tail(const DkfArgs& d, int h, int wq) {
    static_assert(GEO::UPW == 1 && WREG == GEO::TPW && GEO::TPW == 4, "dk tail: one unit per workgroup, sums in registers");
    const Unit un = unit_of(0, wq);
    B::barrier();              // the last pair's output rows have left the exchange buffer
    InnerRegs R;
    BD::template load_inner<false>(R, un);
    dk_tail_tile<0>(d, un, R); dk_tail_tile<1>(d, un, R); dk_tail_tile<2>(d, un, R); dk_tail_tile<3>(d, un, R);
    B::barrier();
    BD::template outer_stage<false, false>(d.Lk, un);
    B::lds_fence();
    dk_tail_rows(d, h, un);
  }
  // Multi-pass sizes (round 5; fft 65536 / 131072, bf16 plans, one chunk per head): at the end of pass k0's pair loop the accumulation
  // registers hold the WHOLE dk_f of the rows (head, k0), so the pass's share of dk -- what Modes::dkifft computes from the fp32 slab
  // of that (head, pass) -- is inverted right here: tile_inv with the pass's twiddle phase, phase C with the pass's inverse outer-digit
  // matrix, and dk_rows_out_rp adds it to the fp32 dk rows (the same wave wrote the earlier passes' sums: its own column slice).
  // No slab (config L = 32K: 402 MB written and read back), no dk_f -> dk launch.
  template <int T>
  static FFC_FN void dk_tail_tile_rp(const DkfArgs& d, Unit un, const InnerRegs& R, Pass ps) {
    A16 re, im;
    w_acc_read<T, 0, 16>(d.dk_scale, re, im);
    BD::template tile_inv<false, true>(d.c.s_inv, un.wq * GEO::TPW + T, R, un, re, im, 0, ps);
  }
  static FFC_FN void dk_tail_rp(const DkfArgs& d, int h, int wq, Pass ps) {
    static_assert(GEO::UPW == 1 && WREG == GEO::TPW && GEO::TPW == 4 && DT == DT_BF16, "dk tail of a pass: one unit per workgroup, bf16 tables");
    const Unit un = unit_of(0, wq);
    B::barrier();              // the last pair's output rows have left the exchange buffer
    InnerRegs R;
    BD::template load_inner<false>(R, un);
    dk_tail_tile_rp<0>(d, un, R, ps); dk_tail_tile_rp<1>(d, un, R, ps); dk_tail_tile_rp<2>(d, un, R, ps); dk_tail_tile_rp<3>(d, un, R, ps);
    B::barrier();
    BD::template outer_stage<false, false, true>(d.Lk, un, 1.0f, ps);
    B::lds_fence();
    DkArgs ka{};
    ka.dk = d.dk_out; ka.H = d.c.H; ka.Lk = d.Lk; ka.fast = d.dk_fast; ka.R = d.c.R;
    dk_rows_out_rp(ka, h, un, ps);
    // (no barrier: the next pass starts with row loads and phase A inside the wave's own column slice, like the next pair of a pass)
  }
  // Several units per workgroup (fft 4096 / 8192 / 16384: UPW pairs of the same head, each with its own sums): tile by tile every
  // wave parks its 32 accumulators in the upper half of the (idle) exchange buffers, unit 0's waves add the units up in a fixed
  // order -- straight into accumulator-shaped registers -- and invert the tile into unit 0's buffer, which lies in the lower half.
  template <int T>
  static FFC_FN void dk_tail_tile_multi(const DkfArgs& d, int u, Unit un, const InnerRegs& R, i32 lane) {
    constexpr int PARK = GEO::L_E + 65536;
    static_assert(GEO::EBYTES <= 65536 && GEO::UPW * GEO::NW * 8192 == 65536, "parking area = the upper half of the exchange buffers");
    w_acc_park<T, 0, 32>(PARK + (u * GEO::NW + un.wq) * 8192 + lane * 4);
    B::barrier();
    if (u == 0) {
      A16 re, im;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        f32 sr = B::fconst(0.f), si = B::fconst(0.f);
#pragma unroll
        for (int s2 = 0; s2 < GEO::UPW; s2++) {
          const i32 src = lane * 4 + (PARK + (s2 * GEO::NW + un.wq) * 8192);
          sr = sr + B::as_f32(B::lds_r32(src + r * 256));
          si = si + B::as_f32(B::lds_r32(src + (16 + r) * 256));
        }
        re[r] = sr * d.dk_scale; im[r] = si * d.dk_scale;
      }
      BD::template tile_inv<false>(d.c.s_inv, un.wq * GEO::TPW + T, R, un, re, im);
    }
    B::barrier();
  }
  static FFC_FN void dk_tail_multi(const DkfArgs& d, int h, int u, int wq) {
    static_assert(GEO::UPW > 1 && WREG == GEO::TPW && GEO::TPW == 4, "dk tail: sums in registers");
    const Unit un = unit_of(u, wq);
    const i32 lane = B::opaque(B::lane());
    B::barrier();              // every unit's last output rows have left the exchange buffers
    InnerRegs R;
    if (u == 0) BD::template load_inner<false>(R, un);
    dk_tail_tile_multi<0>(d, u, un, R, lane); dk_tail_tile_multi<1>(d, u, un, R, lane);
    dk_tail_tile_multi<2>(d, u, un, R, lane); dk_tail_tile_multi<3>(d, u, un, R, lane);
    if (u == 0) {
      // (one wave per unit, fft 4096: its tiles and its column slice are the whole unit -- program order is enough, as in the pair loop)
      BD::template outer_stage<false, false>(d.Lk, un);
      B::lds_fence();
      dk_tail_rows(d, h, un);
    }
    B::barrier();              // persistent kernels (fft 4096): the next job's rows may overwrite the buffers
  }
  // second phase B of dkf / bwd: the tile loop stays rolled (an unrolled one lets the compiler merge the tiles
  // and spill); the resident accumulator of tile slot tt is selected by a wave-uniform switch so that the
  // accumulator registers are addressed statically.
  // (round 5 measured two re-orderings of this loop's loads -- the next tile's spectrum requested early, k_f behind the transform -- both +-0,
  // profiles/r05_ab_zprefetch.txt / r05_ab_kf_late.txt; the switches were removed in round 6)
  template <bool WITH_DX, bool RP = false, bool ZSAVED = false>
  static FFC_FN void bwd_tiles(const ConvArgs& a, int h, Unit un, const InnerRegs& R, const void* zs, float* slab, bool first, WAcc& W,
                               Pass ps = Pass(), bool z_stream = false, bool second = false) {
    // folded outer twiddle (Body::tile_fwd / tile_inv <.., FOLD>): the saved-spectra backward of single-pass fft 32768, whose phase A ran
    // without the twiddle (Modes::bwd)
    constexpr bool FOLD = BD::CAN_FOLD && GEO::N1 == 32 && WITH_DX && !RP && ZSAVED && (WREG >= GEO::TPW);
    const uint8_t* fold = FOLD ? a.tab + a.t.fold : nullptr;
    typename BD::KfRegs zv;
    // FOLD: every matrix is requested one stage ahead of its use; the next tile's first matrix behind this tile's last stage (16 loop-carried registers)
    typename BD::Mat2 fa;
    if constexpr (FOLD) BD::load_mat2_issue(fa, fold + (un.wq * GEO::TPW) * 6144, B::opaque(B::lane()));
#pragma unroll 1
    for (int tt = 0; tt < GEO::TPW; tt++) {
      const int tau = un.wq * GEO::TPW + tt;
#if !defined(FFC_NO_PRIO)
      if constexpr (WITH_DX && GEO::NW > 1) {        // second half of the tile loop: the wave that is behind outranks its partner (bwd)
        if (tt == GEO::TPW / 2) { if (second) B::template setprio<2>(); else B::template setprio<1>(); }
      }
#endif
      z_load(zs, tau, zv, z_stream || (a.flags & 4) != 0);
      typename BD::KfRegs kf;
      constexpr bool KFL = FOLD;      // (folded-twiddle variant: the k_f tile requested behind the transform, its matrices take the registers)
      if constexpr (WITH_DX && !KFL) BD::load_kf(a, h, tau, kf);
      A16 re, im;
      if (WREG >= GEO::TPW || tt < WREG) {
        BD::template tile_fwd<false, false, FOLD>(tau, R, un, re, im, nullptr, fold, FOLD ? &fa : nullptr);
        if constexpr (WITH_DX && KFL) BD::load_kf(a, h, tau, kf);
        switch (tt) {
          case 0: if constexpr (WREG > 0) w_acc_tile<0>(zv, re, im); break;
          case 1: if constexpr (WREG > 1) w_acc_tile<1>(zv, re, im); break;
          case 2: if constexpr (WREG > 2) w_acc_tile<2>(zv, re, im); break;
          default: if constexpr (WREG > 3) w_acc_tile<3>(zv, re, im); break;
        }
      } else {
        WOld wold;
        w_load_old(slab, tau, first, wold);
        BD::template tile_fwd<false>(tau, R, un, re, im);
        if constexpr (WITH_DX && KFL) BD::load_kf(a, h, tau, kf);
        w_update(slab, tau, wold, zv, re, im);
      }
      if constexpr (WITH_DX) {
        typename BD::Mat2 g[2];
        if constexpr (FOLD) {        // the inverse half's matrices, in flight under the accumulation and the k_f product
          const i32 lane = B::opaque(B::lane());
          BD::load_mat2_issue(g[0], fold + (2 * GEO::NT + tau) * 6144, lane);
          BD::load_mat2_issue(g[1], fold + (3 * GEO::NT + tau) * 6144, lane);
        }
        kf_conj_mul(kf, re, im);
        if constexpr (FOLD) {        // next tile's first matrix (clamped on the last iteration)
          const int tn = tt + 1 < GEO::TPW ? tau + 1 : tau;
          BD::load_mat2_issue(fa, fold + tn * 6144, B::opaque(B::lane()));
        }
        BD::template tile_inv<false, RP, false, FOLD>(a.s_inv, tau, R, un, re, im, 0, ps, nullptr, fold, FOLD ? g : nullptr);
      }
    }
  }
  template <bool HALF = false, bool RP = false>
  static FFC_FN void dkf(const DkfArgs& d, int h, int chunk, int wg_linear, int k0 = 0, int wv_in = 0) {
    const ConvArgs& a = d.c;
    const Pass ps = RP ? make_pass(a.tab, a.t, a.R, k0) : Pass();
    constexpr int NCX = HALF ? BD::NCH / 2 : BD::NCH;
    // multi-pass kernels copy the tables and read the wave index once, before their pass loop (nothing derived from the
    // work-item id stays live across the passes: on the 128-VGPR budget it would be parked in an accumulation register)
    if constexpr (!RP) BD::setup_tables(a.tab, a.t);
    const int wv = RP ? wv_in : B::wave();
    Unit un;
    un.wq = wv % GEO::NW;
    const int u = wv / GEO::NW;
    un.eb = u * GEO::EBYTES;
    const int p0 = chunk * a.ppc;
    int p1 = p0 + a.ppc;
    if (p1 > a.npair) p1 = a.npair;
    ConvArgs av = a;            // v = u * pregate
    ConvArgs ad = a;            // dc = dout * postgate
    ad.u = d.dout; ad.pregate = a.postgate; ad.sbu = d.sbd; ad.sbg = a.sbp;
    // OUTER geometries: one slab per chunk (the units are reduced inside the workgroup); else one per (chunk, unit).
    // Multi-pass sizes: slab rows are (head, pass).
    float* slab = RP ? d.ws + (((int64_t)chunk * a.H + h) * a.R + k0) * (GEO::NT * 2048)
                     : d.ws + ((int64_t)chunk * a.H + h) * (GEO::NT * 2048);
    InnerRegs R;
    if constexpr (GEO::OUTER) {
      const int iters = (p1 - p0 + GEO::UPW - 1) / GEO::UPW;
      uint8_t* zs = (uint8_t*)d.zscratch + ((int64_t)(wg_linear * GEO::UPW + u)) * (GEO::N * 4);
      WAcc W;
      w_acc_zero(W);
#pragma unroll 1
      for (int it = 0; it < iters; it++) {
        const int p = p0 + it * GEO::UPW + u;
        const bool act = p < p1;
        if (act) {
          if constexpr (RP) BD::template rows_in_rp<NCX>(av, h, p, un, ps);
          else BD::template rows_in<NCX>(av, h, p, un);
          B::lds_fence();
          BD::template outer_stage<true, HALF, RP>(a.L, un, a.s_fwd, ps);
        }
        BD::unit_barrier();
        if (act) {
          BD::template load_inner<false>(R, un);
#pragma unroll 1
          for (int tt = 0; tt < GEO::TPW; tt++) {
            A16 re, im;
            BD::template tile_fwd<false>(un.wq * GEO::TPW + tt, R, un, re, im);
            z_store(zs, un.wq * GEO::TPW + tt, re, im, (a.flags & 4) != 0);
          }
        }
        BD::unit_barrier();
        if (act) {
          if constexpr (RP) BD::template rows_in_rp<NCX>(ad, h, p, un, ps);
          else BD::template rows_in<NCX>(ad, h, p, un);
          B::lds_fence();
          BD::template outer_stage<true, HALF, RP>(a.L, un, a.s_fwd, ps);
        }
        BD::unit_barrier();
        if (act) {
          BD::template load_inner<false>(R, un);
          bwd_tiles<false, RP>(a, h, un, R, zs, slab, it == 0, W, ps);
        } else if (it == 0) {
#pragma unroll 1
          for (int tt = WREG; tt < GEO::TPW; tt++) w_zero(slab, un.wq * GEO::TPW + tt);
        }
        BD::unit_barrier();
      }
      w_acc_finish(slab, u, un, W);
    } else if (a.R > 1) {
      // inner-only multi-pass form: pass-major (one fp32 W tile in registers per pass), slab rows (head, k0)
      if constexpr (GEO::N == 1024) {
        BD::setup_tables_ipass(a.tab, a.t, a.R);
        const int q0 = p0, q1 = p1;
        const int iters = (q1 - q0 + GEO::UPW - 1) / GEO::UPW;
        BD::template load_inner<false>(R, un);      // (the single-pass twiddle table is not used by the pass forms)
#pragma unroll 1
        for (int k0 = 0; k0 < a.R; k0++) {
          Pass ps; ps.k0 = k0; ps.R = a.R;
          InnerPass ip;
          if constexpr (FFC_IP_LEAN != 0) BD::load_inner_pass_lean(ip, k0);     // round 6: matrices / twiddles of the pass from LDS at their use
          else BD::load_inner_pass(ip, k0);
          A16 wre = B::a16_zero(), wim = B::a16_zero();
#pragma unroll 1
          for (int it = 0; it < iters; it++) {
            const int q = q0 + it * GEO::UPW + u;
            if (q < q1) {
              ZReg zv;
              A16 re, im;
              BD::template rows_in_rp<BD::NCH>(av, h, q, un, ps);
              B::lds_fence();
              BD::template tile_fwd<true, true, false, FFC_IP_LEAN != 0>(0, R, un, re, im, &ip);
              z_pack(re, im, zv);
              B::lds_fence();
              BD::template rows_in_rp<BD::NCH>(ad, h, q, un, ps);
              B::lds_fence();
              BD::template tile_fwd<true, true, false, FFC_IP_LEAN != 0>(0, R, un, re, im, &ip);
              w_add(wre, wim, zv, re, im);
              B::lds_fence();
            }
          }
          reduce_store_w(d.ws + (((int64_t)chunk * a.H + h) * a.R + k0) * 2048, u, wre, wim);
        }
      }
    } else {
      const int q0 = p0 / GEO::G, q1 = (p1 + GEO::G - 1) / GEO::G;
      const int iters = (q1 - q0 + GEO::UPW - 1) / GEO::UPW;
      BD::load_inner(R, un);
      A16 wre, wim;
      wre = B::a16_zero(); wim = B::a16_zero();
#pragma unroll 1
      for (int it = 0; it < iters; it++) {
        const int q = q0 + it * GEO::UPW + u;
        if (q < q1) {
          ZReg zv;
          A16 re, im;
          BD::rows_in(av, h, q, un);
          B::lds_fence();
          BD::tile_fwd(0, R, un, re, im);
          z_pack(re, im, zv);
          B::lds_fence();
          BD::rows_in(ad, h, q, un);
          B::lds_fence();
          BD::tile_fwd(0, R, un, re, im);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            u32 pr = zv.r[r >> 1], pi = zv.i[r >> 1];
            f32 ur = (r & 1) ? B::template unpack_hi<DT>(pr) : B::template unpack_lo<DT>(pr);
            f32 ui = (r & 1) ? B::template unpack_hi<DT>(pi) : B::template unpack_lo<DT>(pi);
            wre[r] = wre[r] + (re[r] * ur + im[r] * ui);
            wim[r] = wim[r] + (im[r] * ur - re[r] * ui);
          }
          B::lds_fence();
        }
      }
      reduce_store_w(slab, u, wre, wim);
    }
  }
  // ------------------------------------------------------------------ fused backward
  // per pair: Z_v = FFT(u*pregate); Z_d = FFT(dout*postgate); W += Z_d conj(Z_v);
  //           dv = iFFT(Z_d conj(k_f)); du = dv * pregate; dpre = dv * u.
  // (reference: kernels_bf16/monarch_cuda_*_bwd_kernel_bf16.h compute the same three transforms)
  static FFC_FN void z_unpack(const typename BD::KfRegs& z, A16& re, A16& im) {
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      u32 wv[4] = {z.v[rq].x, z.v[rq].y, z.v[rq].z, z.v[rq].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        re[4 * rq + q] = B::template unpack_lo<DT>(wv[q]);
        im[4 * rq + q] = B::template unpack_hi<DT>(wv[q]);
      }
    }
  }
  static FFC_FN void kf_plain_mul(const typename BD::KfRegs& kf, A16& re, A16& im) {
    typename BD::CT16 k;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      u32 wv[4] = {kf.v[rq].x, kf.v[rq].y, kf.v[rq].z, kf.v[rq].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        k.re[4 * rq + q] = B::template unpack_lo<DT>(wv[q]);
        k.im[4 * rq + q] = B::template unpack_hi<DT>(wv[q]);
      }
    }
    BD::cmul(re, im, k);
  }
  static FFC_FN void kf_conj_mul(const typename BD::KfRegs& kf, A16& re, A16& im) {
    typename BD::CT16 k;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      u32 wv[4] = {kf.v[rq].x, kf.v[rq].y, kf.v[rq].z, kf.v[rq].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        k.re[4 * rq + q] = B::template unpack_lo<DT>(wv[q]);
        k.im[4 * rq + q] = B::template unpack_hi<DT>(wv[q]);
      }
    }
    BD::cmul_conj(re, im, k);
  }
  // SETUP = false: the plan tables are already in LDS (persistent workgroups of the single-tile sizes copy them once)
  // ZM: 1 = on the spectra the forward pass saved (d.zin), 0 = recomputing, -1 = decided at run time (single-tile kernels).
  // The fused sizes >= 4096 compile the two forms as separate kernels (bwd_kernel<.., ZM>: ffc_k_bwd.hip / ffc_k_bwdz.hip):
  // each carries only its own half of the pair loop, and a profile names them apart.
  template <bool HALF = false, bool RP = false, bool SETUP = true, int ZM = -1>
  static FFC_FN void bwd(const DkfArgs& d, int h, int chunk, int wg_linear, int k0 = 0, int wv_in = 0) {
    const ConvArgs& a = d.c;
    const Pass ps = RP ? make_pass(a.tab, a.t, a.R, k0) : Pass();
    const int hk = RP ? h * a.R + k0 : h;          // k_f row of this (head, pass)
    constexpr int NCX = HALF ? BD::NCH / 2 : BD::NCH;
    // multi-pass kernels copy the tables and read the wave index once, before their pass loop (nothing derived from the
    // work-item id stays live across the passes: on the 128-VGPR budget it would be parked in an accumulation register)
    if constexpr (!RP && SETUP) BD::setup_tables(a.tab, a.t);
    const int wv = RP ? wv_in : B::wave();
    Unit un;
    un.wq = wv % GEO::NW;
    const int u = wv / GEO::NW;
    un.eb = u * GEO::EBYTES;
    const int p0 = chunk * a.ppc;
    int p1 = p0 + a.ppc;
    if (p1 > a.npair) p1 = a.npair;
    ConvArgs av = a;            // v = u * pregate
    ConvArgs ad = a;            // dc = dout * postgate
    ad.u = d.dout; ad.pregate = a.postgate; ad.sbu = d.sbd; ad.sbg = a.sbp;
    ConvArgs ao = a;            // du = dv * pregate
    ao.y = d.du; ao.postgate = a.pregate; ao.sby = d.sbdu; ao.sbp = a.sbg;
    ConvArgs ap = a;            // dpre = dv * u
    ap.y = d.dpre; ap.postgate = a.u; ap.sby = d.sbdpre; ap.sbp = a.sbu;
    ConvArgs aq = a;            // dpost = conv(u*pregate, k) * dout
    aq.y = d.dpost; aq.postgate = d.dout; aq.sby = d.sbdpost; aq.sbp = d.sbd;
    // saved forward output (d.yraw): dpost = dout * yraw falls out of the dout row load, no transform for it
    const bool dpost_tf = d.dpost != nullptr && d.yraw == nullptr;
    // (not in the multi-pass kernels of the fused 32768 size: their 128-VGPR budget has no room for it, build.py check_agpr;
    // the launcher runs the product as a streaming kernel of its own there, ffc_k_bwd.hip mul_rows_kernel)
    if constexpr (!(RP && GEO::OUTER)) {
      if (d.dpost && d.yraw) { ad.aux_in = d.yraw; ad.aux_out = d.dpost; ad.sbai = (int64_t)a.H * a.L; ad.sbao = d.sbdpost; }
    }
    // OUTER geometries: one slab per chunk (the units are reduced inside the workgroup); else one per (chunk, unit).
    // Multi-pass sizes: slab rows are (head, pass).
    float* slab = RP ? d.ws + (((int64_t)chunk * a.H + h) * a.R + k0) * (GEO::NT * 2048)
                     : d.ws + ((int64_t)chunk * a.H + h) * (GEO::NT * 2048);
    InnerRegs R;
    if constexpr (GEO::OUTER) {
      const int iters = (p1 - p0 + GEO::UPW - 1) / GEO::UPW;
      uint8_t* zs = (uint8_t*)d.zscratch + ((int64_t)(wg_linear * GEO::UPW + u)) * (GEO::N * 4);
      WAcc W;
      w_acc_zero(W);
      // profiling build (-DFFC_BWD_PROF, lib/variants): s_memtime sums per phase, [wg][wave][16] in a.prof
#if defined(FFC_BWD_PROF)
      unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = 0, pt1 = 0;
#define FFC_BTICK(k) if constexpr (!RP) { pt1 = B::clock(); pacc[k] += pt1 - pt0; pt0 = pt1; }
#else
#define FFC_BTICK(k)
#endif
      // (round 3: requesting the next rows ahead of the store bursts -- the next pair's u rows before rows_out, the dout rows
      // before the last scratch store -- needs 32 live registers at points where the allocator, which treats a0..a127 as free
      // space, parks them in the accumulation registers; warm-up loads into 4 registers did not shorten the wait either:
      // DESIGN.md section 7)
      const bool have_z = ZM < 0 ? d.zin != nullptr : ZM == 1;
      // FFC_FOLD_TW: the saved-spectra kernel of single-pass fft 32768 runs phase A of dout without the outer twiddle (bwd_tiles folds it)
      constexpr bool FOLDZ = BD::CAN_FOLD && GEO::N1 == 32 && ZM == 1 && !RP && (WREG >= GEO::TPW);
      // input rows of dout by LDS-DMA into the dead half of the exchange buffer (Body::rows_dma): saved-spectra form of the
      // HALF kernels with a 32-point outer digit, plain rows (no gate multiply / side product on the way in), 16-byte-aligned
      // tensors; tuning flag 8 (FFC_FLAGS) keeps the register path for A/B runs
      constexpr bool DMA_OK = BD::HAS_DMA && HALF && !RP && ZM != 0;
      const bool dma = DMA_OK && have_z && a.fast && !ad.pregate && !ad.aux_in && !dpost_tf && !(a.flags & 8);
      if constexpr (DMA_OK) { if (dma && p0 + u < p1) BD::rows_dma(ad, h, p0 + u, un); }
      // wave priority by progress between two barriers (Body::outer_jobs has the measurements): row loads 1, phase A 0, the
      // tile loops 3 then 1 (2 for the second-dispatched wave of the SIMD), phase C 3, stores 2
      const bool second = wv >= 4;
#if defined(FFC_NO_PRIO)
#define FFC_BPRIO(x)
#else
#define FFC_BPRIO(x) if constexpr (GEO::NW > 1) B::template setprio<x>();
#endif
#pragma unroll 1
      for (int it = 0; it < iters; it++) {
        const int p = p0 + it * GEO::UPW + u;
        const bool act = p < p1;
#if defined(FFC_BWD_PROF)
        if constexpr (!RP) pt0 = B::clock();
#endif
        FFC_BPRIO(1)
        // saved spectra (d.zin): the pair's first transform is skipped, its spectrum is read from the forward pass's copy
        const void* zp = !have_z ? (const void*)zs
                         : RP ? (const void*)BD::z_slot_rp(const_cast<void*>(d.zin), h, a.npair, act ? p : p0, a.R, k0)
                              : (const void*)BD::z_slot(const_cast<void*>(d.zin), h, a.npair, act ? p : p0);
        if (have_z) {
          if (dpost_tf) {
            // forward output of this pair for dpost: iFFT(Z_u * k_f) into E.  The previous pair's phase C read every E row.
            BD::unit_barrier();
            if (act) {
              BD::template load_inner<false>(R, un);
#pragma unroll 1
              for (int tt = 0; tt < GEO::TPW; tt++) {
                const int tau = un.wq * GEO::TPW + tt;
                typename BD::KfRegs zv, kf;
                z_load(zp, tau, zv, FFC_Z_STREAM);
                BD::load_kf(a, hk, tau, kf);
                A16 re, im;
                z_unpack(zv, re, im);
                kf_plain_mul(kf, re, im);
                BD::template tile_inv<false, RP>(a.s_inv, tau, R, un, re, im, 0, ps);
              }
            }
            BD::unit_barrier();
            if (act) {
              BD::template outer_stage<false, HALF, RP>(a.L, un, 1.0f, ps);
              B::lds_fence();
              if constexpr (RP) BD::template rows_out_rp<NCX>(aq, h, p, un, ps);
              else BD::template rows_out<NCX>(aq, h, p, un);
            }
          }
          FFC_BTICK(5)
          if (act) {
            bool done = false;
            if constexpr (DMA_OK) {
              if (dma) {      // the rows were requested behind the previous pair's phase C (or in the prologue)
                BD::rows_dma_finish(ad, p, un);
                FFC_BTICK(6)
                FFC_BPRIO(0)
                BD::template outer_stage<true, HALF, RP, true, FOLDZ>(a.L, un, a.s_fwd, ps);
                done = true;
              }
            }
            if (!done) {
              if constexpr (RP) BD::template rows_in_rp<NCX>(ad, h, p, un, ps);
              else BD::template rows_in<NCX>(ad, h, p, un);
              B::lds_fence();
              FFC_BTICK(6)
              FFC_BPRIO(0)
              BD::template outer_stage<true, HALF, RP, false, FOLDZ>(a.L, un, a.s_fwd, ps);
            }
            FFC_BTICK(7)
          }
        } else {
          if (act) {
            if constexpr (RP) BD::template rows_in_rp<NCX>(av, h, p, un, ps);
            else BD::template rows_in<NCX>(av, h, p, un);
            B::lds_fence();
            FFC_BTICK(0)
            FFC_BPRIO(0)
            BD::template outer_stage<true, HALF, RP>(a.L, un, a.s_fwd, ps);
            FFC_BTICK(1)
          }
          BD::unit_barrier();
          FFC_BTICK(2)
          FFC_BPRIO(3)
          if (act) {
            BD::template load_inner<false>(R, un);
#pragma unroll 1
            for (int tt = 0; tt < GEO::TPW; tt++) {
              const int tau = un.wq * GEO::TPW + tt;
              A16 re, im;
              BD::template tile_fwd<false>(tau, R, un, re, im);
              z_store(zs, tau, re, im, (a.flags & 4) != 0);
              if (dpost_tf) {          // forward output of this pair: iFFT(Z_u * k_f) back into E
                // (k_f is requested only now: held across tile_fwd it overflows the 128-VGPR budget into a0..a127)
                typename BD::KfRegs kf;
                BD::load_kf(a, hk, tau, kf);
                kf_plain_mul(kf, re, im);
                BD::template tile_inv<false, RP>(a.s_inv, tau, R, un, re, im, 0, ps);
              }
            }
          }
          FFC_BTICK(3)
          BD::unit_barrier();
          FFC_BTICK(4)
          FFC_BPRIO(1)
          if (dpost_tf) {
            if (act) {
              BD::template outer_stage<false, HALF, RP>(a.L, un, 1.0f, ps);
              B::lds_fence();
              if constexpr (RP) BD::template rows_out_rp<NCX>(aq, h, p, un, ps);    // dpost = y * dout
              else BD::template rows_out<NCX>(aq, h, p, un);
            }
            // no barrier: phase C, rows_out and the rows_in / phase A that follow all stay inside the wave's own
            // column slice of E (same as between two pairs of the forward kernel)
          }
          FFC_BTICK(5)
          if (act) {
            if constexpr (RP) BD::template rows_in_rp<NCX>(ad, h, p, un, ps);
            else BD::template rows_in<NCX>(ad, h, p, un);
            B::lds_fence();
            FFC_BTICK(6)
            FFC_BPRIO(0)
            BD::template outer_stage<true, HALF, RP>(a.L, un, a.s_fwd, ps);
            FFC_BTICK(7)
          }
        }
        BD::unit_barrier();
        FFC_BTICK(8)
        FFC_BPRIO(3)
        if (act) {
          BD::template load_inner<false>(R, un);
          bwd_tiles<true, RP, ZM == 1>(a, hk, un, R, zp, slab, it == 0, W, ps, have_z && FFC_Z_STREAM, second);
        } else if (it == 0) {
#pragma unroll 1
          for (int tt = WREG; tt < GEO::TPW; tt++) w_zero(slab, un.wq * GEO::TPW + tt);
        }
        FFC_BTICK(9)
        BD::unit_barrier();
        FFC_BTICK(10)
        FFC_BPRIO(3)
        if (act) {
          BD::template outer_stage<false, HALF, RP>(a.L, un, 1.0f, ps);
          B::lds_fence();
          FFC_BTICK(11)
          // the next pair's dout rows into E rows 16.. (dead: phase C has read them), in flight under the du stores below
          if constexpr (DMA_OK) { if (dma && p + GEO::UPW < p1) BD::rows_dma(ad, h, p + GEO::UPW, un); }
          FFC_BPRIO(2)
          if constexpr (RP) {
            BD::template rows_out_rp<NCX>(ao, h, p, un, ps);
            if (d.dpre) BD::template rows_out_rp<NCX>(ap, h, p, un, ps);
          } else {
            BD::template rows_out<NCX>(ao, h, p, un);
            if (d.dpre) BD::template rows_out<NCX>(ap, h, p, un);
          }
          FFC_BTICK(12)
        }
      }
#if defined(FFC_BWD_PROF)
      if (!RP && a.prof) {
        const i32 lane = B::lane();
        unsigned long long* dst = a.prof + ((long long)wg_linear * GEO::WGW + wv) * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) B::g_w64(dst, lane * 0 + k, B::u2_from64(pacc[k]), lane < 1);
      }
#endif
#undef FFC_BTICK
#undef FFC_BPRIO
      // (not the one-wave-per-unit kernel of fft 4096: with the tail its register allocation overflows into the accumulation
      // registers, build.py check_agpr)
      if constexpr (!RP && WREG == GEO::TPW && GEO::NW > 1) {
        if (d.dk_out || d.dk_pair) {      // dk straight from the accumulation registers (nchunk == 1)
          using MB = Modes<B, GEO, DT_BF16>;       // the dk inverse always runs in bf16 operand arithmetic
          if constexpr (DT != DT_BF16) {           // fp16 plan: swap the plan's bf16 tables into LDS first
            B::barrier();
            MB::BD::setup_tables(d.tab_bf, d.t_bf);
          }
          if constexpr (GEO::UPW == 1) MB::dk_tail(d, h, un.wq);
          else MB::dk_tail_multi(d, h, u, un.wq);
          return;
        }
      }
      if constexpr (RP && WREG == GEO::TPW && GEO::NW > 1 && GEO::UPW == 1 && DT == DT_BF16) {
        if (d.dk_out) { dk_tail_rp(d, h, un.wq, ps); return; }      // this pass's share of dk from the accumulation registers (nchunk == 1)
      }
      w_acc_finish(slab, u, un, W);
    } else if (a.R > 1) {
      // inner-only multi-pass form: pass-major; du / dpregate accumulate over the passes (rows_out_rp)
      if constexpr (GEO::N == 1024) {
        if constexpr (SETUP) BD::setup_tables_ipass(a.tab, a.t, a.R);
        const int q0 = p0, q1 = p1;
        const int iters = (q1 - q0 + GEO::UPW - 1) / GEO::UPW;
        BD::template load_inner<false>(R, un);      // (the single-pass twiddle table is not used by the pass forms)
#pragma unroll 1
        for (int k0 = 0; k0 < a.R; k0++) {
          Pass ps; ps.k0 = k0; ps.R = a.R;
          InnerPass ip;
          // round 6: the pass's two matrices and its twiddle table are read from LDS where they are used (tile_fwd / tile_inv <.., IPL>); with all
          // 80 registers of a pass resident next to the dk_f sums this kernel spilled 34 - 43 registers into scratch memory inside the pair loop
          // (fft 2048 backward 0.084 -> 0.062 ms at B16 H768, profiles/r06_ab_fft2048.txt)
          if constexpr (FFC_IP_LEAN != 0) BD::load_inner_pass_lean(ip, k0);
          else BD::load_inner_pass(ip, k0);
          A16 wre = B::a16_zero(), wim = B::a16_zero();
#pragma unroll 1
          for (int it = 0; it < iters; it++) {
            const int q = q0 + it * GEO::UPW + u;
            if (q < q1) {
              ZReg zv;
              A16 re, im;
              typename BD::KfRegs kf;
              BD::load_kf(a, h * a.R + k0, 0, kf);
              typename BD::KfRegs zk;
              if (d.zin) {       // spectrum saved by the forward pass: the tile's first transform is skipped
                z_load(BD::z_slot_small(const_cast<void*>(d.zin), h, a.npair, q, a.R, k0), 0, zk, FFC_Z_STREAM);
              } else {
                BD::template rows_in_rp<BD::NCH>(av, h, q, un, ps);
                B::lds_fence();
                BD::template tile_fwd<true, true, false, FFC_IP_LEAN != 0>(0, R, un, re, im, &ip);
                z_pack(re, im, zv);
                B::lds_fence();
              }
              if (ad.aux_in && k0 == 0) BD::template rows_aux_rp<BD::NCH>(ad, h, q, un);      // dpost = dout * yraw, once per pair
              BD::template rows_in_rp<BD::NCH>(ad, h, q, un, ps);
              B::lds_fence();
              BD::template tile_fwd<true, true, false, FFC_IP_LEAN != 0>(0, R, un, re, im, &ip);
              if (d.zin) w_add_k(wre, wim, zk, re, im);
              else w_add(wre, wim, zv, re, im);
              kf_conj_mul(kf, re, im);
              B::lds_fence();
              BD::template tile_inv<true, false, true, false, FFC_IP_LEAN != 0>(a.s_inv, 0, R, un, re, im, 0, Pass(), &ip);
              B::lds_fence();
              BD::template rows_out_rp<BD::NCH>(ao, h, q, un, ps);
              if (d.dpre) BD::template rows_out_rp<BD::NCH>(ap, h, q, un, ps);
              B::lds_fence();
            }
          }
          reduce_store_w(d.ws + (((int64_t)chunk * a.H + h) * a.R + k0) * 2048, u, wre, wim);
        }
      }
    } else {
      const int q0 = p0 / GEO::G, q1 = (p1 + GEO::G - 1) / GEO::G;
      const int iters = (q1 - q0 + GEO::UPW - 1) / GEO::UPW;
      BD::load_inner(R, un);
      A16 wre = B::a16_zero(), wim = B::a16_zero();
#pragma unroll 1
      for (int it = 0; it < iters; it++) {
        const int q = q0 + it * GEO::UPW + u;
        if (q < q1) {
          ZReg zv;
          A16 re, im;
          typename BD::KfRegs kf;
          BD::load_kf(a, h, 0, kf);
          typename BD::KfRegs zk;
          if (d.zin) {       // spectrum saved by the forward pass: the tile's first transform is skipped
            z_load(BD::z_slot_small(const_cast<void*>(d.zin), h, a.npair, q, 1, 0), 0, zk, FFC_Z_STREAM);
          } else {
            BD::rows_in(av, h, q, un);
            B::lds_fence();
            BD::tile_fwd(0, R, un, re, im);
            z_pack(re, im, zv);
            B::lds_fence();
          }
          BD::rows_in(ad, h, q, un);
          B::lds_fence();
          BD::tile_fwd(0, R, un, re, im);
          if (d.zin) w_add_k(wre, wim, zk, re, im);
          else w_add(wre, wim, zv, re, im);
          kf_conj_mul(kf, re, im);
          B::lds_fence();
          BD::tile_inv(a.s_inv, 0, R, un, re, im);
          B::lds_fence();
          BD::rows_out(ao, h, q, un);
          if (d.dpre) BD::rows_out(ap, h, q, un);
          B::lds_fence();
        }
      }
      reduce_store_w(slab, u, wre, wim);
    }
  }

  // W += Zd * conj(Zv) (Zv as packed dtype pairs)
  static FFC_FN void w_add(A16& wre, A16& wim, const ZReg& zv, const A16& re, const A16& im) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      u32 pr = zv.r[r >> 1], pi = zv.i[r >> 1];
      f32 ur = (r & 1) ? B::template unpack_hi<DT>(pr) : B::template unpack_lo<DT>(pr);
      f32 ui = (r & 1) ? B::template unpack_hi<DT>(pi) : B::template unpack_lo<DT>(pi);
      wre[r] = wre[r] + (re[r] * ur + im[r] * ui);
      wim[r] = wim[r] + (im[r] * ur - re[r] * ui);
    }
  }
  // the same with Zv in the saved-spectrum format (z_store: one (re, im) dtype pair per word)
  static FFC_FN void w_add_k(A16& wre, A16& wim, const typename BD::KfRegs& z, const A16& re, const A16& im) {
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      u32 wv[4] = {z.v[rq].x, z.v[rq].y, z.v[rq].z, z.v[rq].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 4 * rq + q;
        f32 ur = B::template unpack_lo<DT>(wv[q]), ui = B::template unpack_hi<DT>(wv[q]);
        wre[r] = wre[r] + (re[r] * ur + im[r] * ui);
        wim[r] = wim[r] + (im[r] * ur - re[r] * ui);
      }
    }
  }
  // Single-tile sizes: the eight units' sums of a (head, chunk[, pass]) job are added up through LDS -- the units' exchange
  // buffers, 8 x 4 KB, free between two jobs -- and leave as ONE slab.  (Round 2 wrote one slab per unit: 8x the partial-sum
  // traffic and a serial 8-slab loop in dkifft, 30 us at fft 256 against 22 us for the backward kernel itself.)
  // Wave u ends up with accumulator registers 2u, 2u+1 of every lane = one 16-byte store of store_w's layout.
  static FFC_FN void reduce_store_w(float* slab, int u, const A16& wre, const A16& wim) {
    static_assert(!GEO::OUTER, "single-tile geometries");
    if constexpr (GEO::UPW == 1) {
      store_w(slab, wre, wim);
    } else {
      static_assert(GEO::UPW == 8 && GEO::UPW * GEO::EBYTES >= 8 * 4096, "one wave per unit, 4 KB of exchange buffer each");
      const i32 lane = B::opaque(B::lane());
      f32 sr[2], si[2];
#pragma unroll
      for (int plane = 0; plane < 2; plane++) {
        B::barrier();                 // every unit is done with its exchange buffer (plane 1: with the sums of plane 0)
#pragma unroll
        for (int r = 0; r < 16; r++) B::lds_w32(((u * 16 + r) * 64 + lane) * 4, B::as_u32(plane ? wim[r] : wre[r]));
        B::barrier();
#pragma unroll
        for (int k = 0; k < 2; k++) {
          f32 acc = B::fconst(0.f);
#pragma unroll
          for (int v = 0; v < 8; v++) acc = acc + B::as_f32(B::lds_r32(((v * 16 + (2 * u + k)) * 64 + lane) * 4));
          if (plane) si[k] = acc; else sr[k] = acc;
        }
      }
      B::barrier();                   // the next job's rows may overwrite the buffers
      const i32 c = lane & 31, hi = lane >> 5;
      const i32 idx = ((hi + 2 * (u >> 1)) * 32 + c) * 2 + (u & 1);
      U4 n;
      n.x = B::as_u32(sr[0]); n.y = B::as_u32(si[0]); n.z = B::as_u32(sr[1]); n.w = B::as_u32(si[1]);
      B::g_w128(slab, idx, n, B::ptrue());
    }
  }
  static FFC_FN void store_w(float* slab, const A16& re, const A16& im) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
      i32 idx = ((hi + 2 * rq) * 32 + c) * 2;
      U4 n0, n1;
      n0.x = B::as_u32(re[4 * rq]); n0.y = B::as_u32(im[4 * rq]); n0.z = B::as_u32(re[4 * rq + 1]); n0.w = B::as_u32(im[4 * rq + 1]);
      n1.x = B::as_u32(re[4 * rq + 2]); n1.y = B::as_u32(im[4 * rq + 2]); n1.z = B::as_u32(re[4 * rq + 3]); n1.w = B::as_u32(im[4 * rq + 3]);
      B::g_w128(slab, idx, n0, B::ptrue());
      B::g_w128(slab, idx + 1, n1, B::ptrue());
    }
  }

  // ------------------------------------------------------------------ dk_f -> dk
  // hmul: slab rows are (head, pass) for the multi-pass sizes (unit_id is then head * R + k0)
  static FFC_FN void w_load(const DkArgs& a, int unit_id, int tau, A16& re, A16& im, int hmul = 1) {
    const i32 lane = B::opaque(B::lane());
    const i32 c = lane & 31, hi = lane >> 5;
    const int64_t slab_stride = (int64_t)a.H * hmul * (GEO::NT * 2048);   // floats
    re = B::a16_zero(); im = B::a16_zero();
#pragma unroll 1
    for (int s = 0; s < a.nslab; s++) {
      const float* sl = a.ws + s * slab_stride;
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        if constexpr (GEO::OUTER) {
          i32 idx = ((hi + (tau * 8 + 2 * rq)) * 32 + c) * 2 + unit_id * (GEO::NT * 512);
          U4 o0 = B::g_r128(sl, idx), o1 = B::g_r128(sl, idx + 1);
          re[4 * rq] = re[4 * rq] + B::as_f32(o0.x); im[4 * rq] = im[4 * rq] + B::as_f32(o0.y);
          re[4 * rq + 1] = re[4 * rq + 1] + B::as_f32(o0.z); im[4 * rq + 1] = im[4 * rq + 1] + B::as_f32(o0.w);
          re[4 * rq + 2] = re[4 * rq + 2] + B::as_f32(o1.x); im[4 * rq + 2] = im[4 * rq + 2] + B::as_f32(o1.y);
          re[4 * rq + 3] = re[4 * rq + 3] + B::as_f32(o1.z); im[4 * rq + 3] = im[4 * rq + 3] + B::as_f32(o1.w);
        } else {
          i32 V = hi * 4 + 8 * rq;
          i32 sV = V / GEO::N3, k3 = V % GEO::N3, sU = c / GEO::N2, k2 = c % GEO::N2;
          i32 hd = sU * GEO::SV + sV + unit_id * GEO::G;
          pred ok = hd < a.H * hmul;
#pragma unroll
          for (int su = 0; su < GEO::SU; su++)
#pragma unroll
            for (int sv = 0; sv < GEO::SV; sv++) {
              i32 rho = (k3 + sv * GEO::N3) >> 2;
              i32 idx = (rho * 32 + (k2 + su * GEO::N2)) * 2 + hd * 512;
              U4 o0 = B::g_r128p(sl, idx, ok), o1 = B::g_r128p(sl, idx + 1, ok);
              re[4 * rq] = re[4 * rq] + B::as_f32(o0.x); im[4 * rq] = im[4 * rq] + B::as_f32(o0.y);
              re[4 * rq + 1] = re[4 * rq + 1] + B::as_f32(o0.z); im[4 * rq + 1] = im[4 * rq + 1] + B::as_f32(o0.w);
              re[4 * rq + 2] = re[4 * rq + 2] + B::as_f32(o1.x); im[4 * rq + 2] = im[4 * rq + 2] + B::as_f32(o1.y);
              re[4 * rq + 3] = re[4 * rq + 3] + B::as_f32(o1.z); im[4 * rq + 3] = im[4 * rq + 3] + B::as_f32(o1.w);
            }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) { re[r] = re[r] * a.scale; im[r] = im[r] * a.scale; }
  }
  static FFC_FN void dk_rows_out(const DkArgs& a, int unit_id, Unit un) {
    const i32 lane = B::opaque(B::lane());
#pragma unroll
    for (int i = 0; i < BD::NCH; i++) {
      i32 idx = lane + i * 64;
      i32 row = idx / BD::CPR, m = (idx % BD::CPR) * 8 + (GEO::OUTER ? un.wq * 128 * GEO::S1 : 0);
      pred sw;
      i32 off = BD::pair_off(row, m, &sw) + un.eb;
      U4 o = B::lds_r128(off);
      u32 w[4];
      w[0] = B::sel(sw, o.z, o.x); w[1] = B::sel(sw, o.w, o.y); w[2] = B::sel(sw, o.x, o.z); w[3] = B::sel(sw, o.y, o.w);
      i32 hd, n;
      if constexpr (GEO::OUTER) { hd = row * 0 + unit_id; n = row * GEO::Mi + m; }
      else { hd = row + unit_id * GEO::G; n = m; }
      pred ok = hd < a.H;
      i32 e0 = hd * a.Lk + n;
      f32 v[8];
#pragma unroll
      for (int q = 0; q < 4; q++) { v[2 * q] = B::template unpack_lo<DT>(w[q]); v[2 * q + 1] = B::template unpack_hi<DT>(w[q]); }
      if (a.fast) {
        U4 s0, s1;
        s0.x = B::as_u32(v[0]); s0.y = B::as_u32(v[1]); s0.z = B::as_u32(v[2]); s0.w = B::as_u32(v[3]);
        s1.x = B::as_u32(v[4]); s1.y = B::as_u32(v[5]); s1.z = B::as_u32(v[6]); s1.w = B::as_u32(v[7]);
        B::g_w128(a.dk, e0 >> 2, s0, ok && (n < a.Lk));
        B::g_w128(a.dk, (e0 >> 2) + 1, s1, ok && ((n + 4) < a.Lk));
      } else {
#pragma unroll
        for (int q = 0; q < 8; q++) B::g_w32(a.dk, e0 + q, B::as_u32(v[q]), ok && ((n + q) < a.Lk));
      }
    }
  }
  // multi-pass sizes: dk[n0 M + m] (+)= Re(i^q y_k0[m]) = {r, -s, -r, s}[q], q = n0 k0 4/R; fp32 accumulation over the passes
  static FFC_FN void dk_rows_out_rp(const DkArgs& a, int unit_id, Unit un, Pass ps) {
    const i32 lane = B::opaque(B::lane());
    const int n0max = (a.Lk + GEO::N - 1) / GEO::N;
#pragma unroll
    for (int i = 0; i < BD::NCH; i++) {
      i32 idx = lane + i * 64;
      i32 row = idx / BD::CPR, m = (idx % BD::CPR) * 8 + un.wq * 128 * GEO::S1;
      pred sw;
      i32 off = BD::pair_off(row, m, &sw) + un.eb;
      u32 w[2][4];
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        U4 o = B::lds_r128(off + pl * GEO::PLANE);
        w[pl][0] = B::sel(sw, o.z, o.x); w[pl][1] = B::sel(sw, o.w, o.y); w[pl][2] = B::sel(sw, o.x, o.z); w[pl][3] = B::sel(sw, o.y, o.w);
      }
      i32 hd = row * 0 + unit_id;
      pred ok = hd < a.H;
#pragma unroll 1
      for (int n0 = 0; n0 < n0max; n0++) {
        const int q = (n0 * ps.k0 * (4 / ps.R)) & 3;
        const int pl = q & 1;
        const float sg = (q == 0 || q == 3) ? 1.0f : -1.0f;
        i32 n = row * GEO::Mi + m + n0 * GEO::N;
        i32 e0 = hd * a.Lk + n;
        f32 v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          v[2 * k] = B::template unpack_lo<DT>(w[pl][k]) * sg; v[2 * k + 1] = B::template unpack_hi<DT>(w[pl][k]) * sg;
        }
        if (ps.k0 > 0) {
          f32 old[8];
          fload8(a.dk, e0, n, a.Lk, a.fast != 0, ok, old);
#pragma unroll
          for (int k = 0; k < 8; k++) v[k] = v[k] + old[k];
        }
        if (a.fast) {
          U4 s0, s1;
          s0.x = B::as_u32(v[0]); s0.y = B::as_u32(v[1]); s0.z = B::as_u32(v[2]); s0.w = B::as_u32(v[3]);
          s1.x = B::as_u32(v[4]); s1.y = B::as_u32(v[5]); s1.z = B::as_u32(v[6]); s1.w = B::as_u32(v[7]);
          B::g_w128(a.dk, e0 >> 2, s0, ok && (n < a.Lk));
          B::g_w128(a.dk, (e0 >> 2) + 1, s1, ok && ((n + 4) < a.Lk));
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) B::g_w32(a.dk, e0 + k, B::as_u32(v[k]), ok && ((n + k) < a.Lk));
        }
      }
    }
  }
  static FFC_FN void dkifft(const DkArgs& a, int wg) {
    BD::setup_tables(a.tab, a.t);
    const int wv = B::wave();
    Unit un;
    un.wq = wv % GEO::NW;
    const int u = wv / GEO::NW;
    un.eb = u * GEO::EBYTES;
    const int unit_id = wg * GEO::UPW + u;
    const int nunits = GEO::OUTER ? a.H : (a.H + GEO::G - 1) / GEO::G;
    const bool act = unit_id < nunits;
    InnerRegs R;
    BD::load_inner(R, un);
    if constexpr (GEO::N == 32768) {
      if (a.R > 1) {        // multi-pass size: one inverse per pass, dk accumulated in fp32 by the same wave
#pragma unroll 1
        for (int k0 = 0; k0 < a.R; k0++) {
          const Pass ps = make_pass(a.tab, a.t, a.R, k0);
          if (act) {
#pragma unroll 1
            for (int tt = 0; tt < GEO::TPW; tt++) {
              A16 re, im;
              w_load(a, unit_id * a.R + k0, un.wq * GEO::TPW + tt, re, im, a.R);
              BD::template tile_inv<true, true>(a.s_inv, un.wq * GEO::TPW + tt, R, un, re, im, 0, ps);
            }
          }
          B::barrier();
          if (act) {
            BD::template outer_stage<false, false, true>(a.Lk, un, 1.0f, ps);
            B::lds_fence();
            if (a.outpair) {     // complex output (pair-plane tensor (2, H, R*M) dtype) accumulated over the passes
              ConvArgs cv{};
              cv.y = a.outpair; cv.B = 2; cv.H = a.H; cv.L = GEO::N * a.R; cv.fast = 1; cv.sby = (int64_t)a.H * cv.L;
              BD::template rows_out_rp<BD::NCH>(cv, unit_id, 0, un, ps);
            } else {
              dk_rows_out_rp(a, unit_id, un, ps);
            }
          }
          B::barrier();
        }
        return;
      }
    }
    if constexpr (GEO::OUTER) {
      if (act) {
#pragma unroll 1
        for (int tt = 0; tt < GEO::TPW; tt++) {
          A16 re, im;
          w_load(a, unit_id, un.wq * GEO::TPW + tt, re, im);
          BD::tile_inv(a.s_inv, un.wq * GEO::TPW + tt, R, un, re, im, a.flags);
        }
      }
      B::barrier();
      if (act) {
        BD::template outer_stage<false, false>(a.Lk, un);
        B::lds_fence();
        if (a.outpair) {
          ConvArgs cv{};
          cv.y = a.outpair; cv.B = 2; cv.H = a.H; cv.L = GEO::N; cv.fast = 1; cv.sby = (int64_t)a.H * GEO::N;
          BD::rows_out(cv, unit_id, 0, un);
        } else {
          dk_rows_out(a, unit_id, un);
        }
      }
    } else if (a.R > 1) {
      if constexpr (GEO::N == 1024) {
        BD::setup_tables_ipass(a.tab, a.t, a.R);
        if (act) {
#pragma unroll 1
          for (int k0 = 0; k0 < a.R; k0++) {
            Pass ps; ps.k0 = k0; ps.R = a.R;
            InnerPass ip;
            BD::load_inner_pass(ip, k0);
            A16 re, im;
            w_load(a, unit_id * a.R + k0, 0, re, im, a.R);
            BD::template tile_inv<true, false, true, false, FFC_IP_LEAN != 0>(a.s_inv, 0, R, un, re, im, 0, Pass(), &ip);
            B::lds_fence();
            dk_rows_out_rp(a, unit_id, un, ps);
            B::lds_fence();
          }
        }
      }
    } else {
      if (act) {
        A16 re, im;
        w_load(a, unit_id, 0, re, im);
        BD::tile_inv(a.s_inv, 0, R, un, re, im);
        B::lds_fence();
        dk_rows_out(a, unit_id, un);
      }
    }
  }
};

}  // namespace ffc
