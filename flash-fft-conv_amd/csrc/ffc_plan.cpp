// Host plan builder (pure C++, shared by the HIP library and the CPU wave simulator).
#include "ffc_plan.h"

#include <math.h>
#include <string.h>

#include "ffc_layout.h"

namespace ffc {

uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  uint32_t r = 0x7fffu + ((u >> 16) & 1);
  return (uint16_t)((u + r) >> 16);
}
float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to inf (>= 65520)
  if (ax < 0x33000001u) return (uint16_t)sign;               // rounds to zero (<= 2^-25)
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t base;
  if (e < -14) {  // subnormal half
    shift = 13 + (-14 - e);
    base = 0;
  } else {
    shift = 13;
    base = (uint32_t)(e + 15) << 10;
    m &= 0x7fffffu;
  }
  uint32_t q = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1))) q++;
  return (uint16_t)(sign | (base + q));
}
float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  int e = (h >> 10) & 31;
  uint32_t m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) {
      u = sign;
    } else {
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; sh++; }
      m &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
    }
  } else if (e == 31) {
    u = sign | 0x7f800000u | (m << 13);
  } else {
    u = sign | ((uint32_t)(e - 15 + 127) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

bool plan_factors(int N, int* n1, int* n2, int* n3, int* passes) {
  if (passes) *passes = 1;
  if (N == 65536 || N == 131072) {     // R passes of the 32 x 32 x 32 kernel
    *n1 = 32; *n2 = 32; *n3 = 32;
    if (passes) *passes = N / 32768;
    return true;
  }
  if (N == 2048) {                     // 2 passes of the inner-only 32 x 32 kernel
    *n1 = 1; *n2 = 32; *n3 = 32;
    if (passes) *passes = 2;
    return true;
  }
  switch (N) {
    case 256: *n1 = 1; *n2 = 16; *n3 = 16; return true;
    case 512: *n1 = 1; *n2 = 16; *n3 = 32; return true;
    case 1024: *n1 = 1; *n2 = 32; *n3 = 32; return true;
    case 4096: *n1 = 16; *n2 = 16; *n3 = 16; return true;
    case 8192: *n1 = 32; *n2 = 16; *n3 = 16; return true;
    case 16384: *n1 = 16; *n2 = 32; *n3 = 32; return true;
    case 32768: *n1 = 32; *n2 = 32; *n3 = 32; return true;
  }
  return false;
}

namespace {

const double kPi = 3.14159265358979323846;

struct Builder {
  std::vector<uint8_t>& b;
  int alloc(int bytes) {
    int off = (int)b.size();
    b.resize(off + ((bytes + 255) & ~255), 0);
    return off;
  }
};

uint16_t to_dt(double v, int dtype) {
  return dtype == DT_BF16 ? f32_to_bf16((float)v) : f32_to_f16((float)v);
}

// Operand table of the block-diagonal Nd-point DFT: [q = ms*3 + which][lane][d].
void fill_mat(uint8_t* dst, int Nd, int dtype) {
  uint32_t* w = (uint32_t*)dst;
  for (int ms = 0; ms < 2; ms++)
    for (int which = 0; which < 3; which++)
      for (int lane = 0; lane < 64; lane++)
        for (int d = 0; d < 4; d++) {
          uint32_t word = 0;
          for (int half = 0; half < 2; half++) {
            int e = 2 * d + half;
            int nw = lane & 31, old = kslot_row(ms, lane >> 5, e);
            double re = 0, im = 0;
            if (nw / Nd == old / Nd) {
              int k = nw % Nd, n = old % Nd;
              double ang = -2.0 * kPi * (double)((n * k) % Nd) / Nd;
              re = cos(ang);
              im = sin(ang);
            }
            double v = which == 0 ? re : (which == 1 ? im : -im);
            word |= (uint32_t)to_dt(v, dtype) << (16 * half);
          }
          w[((ms * 3 + which) * 64 + lane) * 4 + d] = word;
        }
}

// K-step-0 operand table of the 32-point DFT for the frequency-sparse inverse stage (PlanTabs::mat_sp): slots e < 4 of lane
// half hi hold the contraction row k3 = e (hi = 0) / 28 + e (hi = 1), slots e >= 4 carry zero data.  [q = which][lane][d]
void fill_mat_sparse(uint8_t* dst, int dtype) {
  uint32_t* w = (uint32_t*)dst;
  for (int which = 0; which < 3; which++)
    for (int lane = 0; lane < 64; lane++)
      for (int d = 0; d < 4; d++) {
        uint32_t word = 0;
        for (int half = 0; half < 2; half++) {
          int e = 2 * d + half;
          int out = lane & 31, con = (lane >> 5) ? 28 + e : e;
          double re = 0, im = 0;
          if (e < 4) {
            double ang = -2.0 * kPi * (double)((out * con) % 32) / 32.0;
            re = cos(ang); im = sin(ang);
          }
          double v = which == 0 ? re : (which == 1 ? im : -im);
          word |= (uint32_t)to_dt(v, dtype) << (16 * half);
        }
        w[(which * 64 + lane) * 4 + d] = word;
      }
}

// Outer-digit operand table of pass k0 of an R-pass size (Nd = 32): the Nd-point DFT times the pass factor
// W_{R Nd}^{n1 k0} on the INPUT index n1.  Same [out][contraction] operand layout as fill_mat.  Forward (phase A):
// out = k1, contraction = n1: F[k1][n1] = W_Nd^{n1 k1} W_{R Nd}^{n1 k0}.  Inverse (phase C, used with the kernels' CONJ
// flag): out = n1, contraction = k1, the factor sits on the OUTPUT index: T[n1][k1] = W_Nd^{n1 k1} W_{R Nd}^{n1 k0}.
void fill_mat_pass(uint8_t* dst, int Nd, int dtype, int k0, int R, bool inverse) {
  uint32_t* w = (uint32_t*)dst;
  for (int ms = 0; ms < 2; ms++)
    for (int which = 0; which < 3; which++)
      for (int lane = 0; lane < 64; lane++)
        for (int d = 0; d < 4; d++) {
          uint32_t word = 0;
          for (int half = 0; half < 2; half++) {
            int e = 2 * d + half;
            int out = (lane & 31) % Nd, con = kslot_row(ms, lane >> 5, e) % Nd;
            int n1 = inverse ? out : con;
            long ph = ((long)out * con * R + (long)n1 * k0) % ((long)R * Nd);
            double ang = -2.0 * kPi * (double)ph / (double)(R * Nd);
            double re = cos(ang), im = sin(ang);
            double v = which == 0 ? re : (which == 1 ? im : -im);
            word |= (uint32_t)to_dt(v, dtype) << (16 * half);
          }
          w[((ms * 3 + which) * 64 + lane) * 4 + d] = word;
        }
}

// Per-tile inner matrices with the outer twiddle folded in (PlanTabs::fold; Geo<32,32,32>, tile = k1).  out = lane's (non-contracted)
// index, con = contraction index; the extra factor exp(-+ 2 pi i x k1 mult / N) sits on x = con (forward stages: input index) or
// x = out (inverse stages: output index); mult = 32 for the n2 stages (m = 32 n2 + n3), 1 for the n3 stages.
void fill_mat_fold(uint8_t* dst, int dtype, int which4, int k1, int N, double scale) {
  uint32_t* w = (uint32_t*)dst;
  const bool inverse = which4 >= 2;
  const int mult = (which4 == 0 || which4 == 3) ? 32 : 1;
  for (int ms = 0; ms < 2; ms++)
    for (int which = 0; which < 3; which++)
      for (int lane = 0; lane < 64; lane++)
        for (int d = 0; d < 4; d++) {
          uint32_t word = 0;
          for (int half = 0; half < 2; half++) {
            int e = 2 * d + half;
            int out = lane & 31, con = kslot_row(ms, lane >> 5, e);
            int x = inverse ? out : con;
            long ph = ((long)out * con * (N / 32) + (long)x * k1 * mult) % N;      // in units of 2 pi / N
            double ang = (inverse ? 2.0 : -2.0) * kPi * (double)ph / (double)N;
            double re = scale * cos(ang), im = scale * sin(ang);
            double v = which == 0 ? re : (which == 1 ? im : -im);
            word |= (uint32_t)to_dt(v, dtype) << (16 * half);
          }
          w[((ms * 3 + which) * 64 + lane) * 4 + d] = word;
        }
}

// ctab16: [rr 8][lane 64][re(2rr) im(2rr) re(2rr+1) im(2rr+1)]
template <class F>
void fill_ctab16(uint8_t* dst, F fn) {
  float* w = (float*)dst;
  for (int lane = 0; lane < 64; lane++)
    for (int r = 0; r < 16; r++) {
      double re, im;
      fn(lane, r, &re, &im);
      int rr = r >> 1, o = r & 1;          // per lane and row pair: (re0, re1, im0, im1) = packed-math operands
      w[(rr * 64 + lane) * 4 + o] = (float)re;
      w[(rr * 64 + lane) * 4 + o + 2] = (float)im;
    }
}

void cis(double num, double den, double scale, double* re, double* im) {
  double ang = 2.0 * kPi * fmod(num, den) / den;
  *re = scale * cos(ang);
  *im = scale * sin(ang);
}

template <class GEO>
void build(HostPlan* p) {
  Builder bl{p->blob};
  const int N = GEO::N * p->R;         // the fft size (R passes of the GEO::N kernel)
  p->NT = GEO::NT; p->NW = GEO::NW; p->G = GEO::G;
  int lg = 0;
  while ((1 << lg) < N) lg++;
  p->s_fwd = ldexp(1.0, -((lg + 1) / 2));
  // forward carries s_fwd ~ 1/sqrt(N) (spectrum RMS ~ input RMS), k_f is stored unscaled and the
  // remaining 1/(N*s_fwd) is applied in fp32 at the last inner inverse twiddle, so that no
  // intermediate drifts into the fp16 subnormal range.
  p->s_k = 1.0;
  p->s_inv = 1.0 / ((double)N * p->s_fwd);
  PlanTabs& t = p->tabs;
  int digits[3] = {GEO::N1, GEO::N2, GEO::N3};
  for (int i = 0; i < 3; i++) {
    t.mat[i] = bl.alloc(6 * 64 * 16);
    if (digits[i] > 1) fill_mat(p->blob.data() + t.mat[i], digits[i], p->dtype);
  }
  const double sf_inner = GEO::OUTER ? 1.0 : p->s_fwd;
  t.twin = bl.alloc(8192);
  fill_ctab16(p->blob.data() + t.twin, [&](int lane, int r, double* re, double* im) {
    int k2 = (lane & 31) % GEO::N2, n3 = acc_row(r, lane >> 5) % GEO::N3;
    cis(-(double)(n3 * k2), GEO::Mi, sf_inner, re, im);
  });
  t.twin2 = bl.alloc(8192);
  fill_ctab16(p->blob.data() + t.twin2, [&](int lane, int r, double* re, double* im) {
    int n3 = (lane & 31) % GEO::N3, k2 = acc_row(r, lane >> 5) % GEO::N2;
    cis((double)(n3 * k2), GEO::Mi, GEO::OUTER ? 1.0 : p->s_inv, re, im);
  });
  for (int k0 = 0; k0 < 4; k0++) t.ipass[k0] = 0;
  t.mat_sp = bl.alloc(3 * 64 * 16);
  fill_mat_sparse(p->blob.data() + t.mat_sp, p->dtype);
  if (!GEO::OUTER && p->R > 1) {
    // inner-only multi-pass form: with m = N3 n2 + n3 the pass factor W_N^{m k0} = W_{N/N3}^{n2 k0} W_N^{n3 k0}; its n2 part
    // multiplies the stage-a matrix (contraction index) and, conjugated by the kernels' CONJ flag, the last inverse matrix
    // (output index); its n3 part multiplies the two inner twiddle tables.
    for (int k0 = 0; k0 < p->R; k0++) {
      t.ipass[k0] = bl.alloc(2 * 6144 + 2 * 8192);
      uint8_t* q = p->blob.data() + t.ipass[k0];
      fill_mat_pass(q, GEO::N2, p->dtype, k0, p->R, false);
      fill_mat_pass(q + 6144, GEO::N2, p->dtype, k0, p->R, true);
      const int R = p->R;
      fill_ctab16(q + 12288, [&](int lane, int r, double* re, double* im) {
        int k2 = (lane & 31) % GEO::N2, n3 = acc_row(r, lane >> 5) % GEO::N3;
        cis(-(double)n3 * (k2 * R + k0), (double)GEO::Mi * R, p->s_fwd, re, im);
      });
      fill_ctab16(q + 12288 + 8192, [&](int lane, int r, double* re, double* im) {
        int n3 = (lane & 31) % GEO::N3, k2 = acc_row(r, lane >> 5) % GEO::N2;
        cis((double)n3 * (k2 * R + k0), (double)GEO::Mi * R, p->s_inv, re, im);
      });
    }
  }
  for (int k0 = 0; k0 < 4; k0++) t.matk[k0][0] = t.matk[k0][1] = t.mat[0];
  for (int k0 = 1; k0 < p->R; k0++)
    for (int inv = 0; inv < 2; inv++) {
      t.matk[k0][inv] = bl.alloc(6 * 64 * 16);
      fill_mat_pass(p->blob.data() + t.matk[k0][inv], GEO::N1, p->dtype, k0, p->R, inv != 0);
    }
  t.fold = 0;
  // (fft 16384: 384 KB per plan, product; fft 32768: 768 KB, only in builds with FFC_FOLD_TW=2 -- measured and not adopted)
  if (((FFC_FOLD_TW >= 1 && GEO::N1 == 16) || (FFC_FOLD_TW >= 2 && GEO::N1 == 32)) && GEO::N2 == 32 && GEO::N3 == 32 && p->R == 1) {
    t.fold = bl.alloc(4 * GEO::NT * 6144);
    for (int which4 = 0; which4 < 4; which4++)
      for (int k1 = 0; k1 < GEO::NT; k1++)
        fill_mat_fold(p->blob.data() + t.fold + (which4 * GEO::NT + k1) * 6144, p->dtype, which4, k1, GEO::N,
                      which4 == 0 ? p->s_fwd : (which4 == 3 ? p->s_inv : 1.0));
  }
  t.total = (int)p->blob.size();
  // internal position -> natural frequency: pass k0 holds f = k0 + R * f' (f' = the inner kernel's frequency)
  p->kf_freq.resize((size_t)p->R * GEO::NT * 1024);
  for (int k0 = 0; k0 < p->R; k0++)
    for (int tau = 0; tau < GEO::NT; tau++)
      for (int rho = 0; rho < 8; rho++)
        for (int U = 0; U < 32; U++)
          for (int v = 0; v < 4; v++)
            p->kf_freq[(((size_t)k0 * GEO::NT + tau) * 8 + rho) * 128 + U * 4 + v] = k0 + p->R * kf_freq<GEO>(tau, 4 * rho + v, U);
}

}  // namespace

bool build_plan(int N, int dtype, HostPlan* p) {
  int n1, n2, n3, passes;
  if (!plan_factors(N, &n1, &n2, &n3, &passes)) return false;
  if (dtype != DT_BF16 && dtype != DT_F16) return false;
  *p = HostPlan();
  p->N = N; p->N1 = n1; p->N2 = n2; p->N3 = n3; p->dtype = dtype; p->R = passes;
  switch (N / passes) {
    case 256: build<Geo<1, 16, 16>>(p); break;
    case 512: build<Geo<1, 16, 32>>(p); break;
    case 1024: build<Geo<1, 32, 32>>(p); break;
    case 4096: build<Geo<16, 16, 16>>(p); break;
    case 8192: build<Geo<32, 16, 16>>(p); break;
    case 16384: build<Geo<16, 32, 32>>(p); break;
    case 32768: build<Geo<32, 32, 32>>(p); break;
    default: return false;
  }
  return true;
}

}  // namespace ffc
