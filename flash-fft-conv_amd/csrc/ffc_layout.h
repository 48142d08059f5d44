// Index algebra shared by the host plan builder, the kernel body and the CPU wave simulator.
//
// Monarch decomposition used here (MI355X design, not the reference's):
//   N = N1*N2*N3,  n = n1*Mi + n2*N3 + n3 (Mi = N2*N3),  f = k1 + N1*(k2 + N2*k3)
//   stage order fwd: n1 (outer, strided) -> n2 -> n3 ; inverse: k3 -> k2 -> k1.
// Every DFT stage is a 32x32 complex tile product on v_mfma_f32_32x32x16_{bf16,f16};
// 16-point digits are embedded block-diagonally (two independent sub-blocks per tile).
// The reference's factor tables are flashfftconv/conv.py:72-551 (semantic equivalent).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FFC_HD __host__ __device__
#else
#define FFC_HD
#endif

namespace ffc {

// v_mfma_f32_32x32x16 C/D layout: lane l (hi = l>>5), register r -> row, col = l&31.
FFC_HD constexpr int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
// Logical contraction row carried by K-step ms, lane half hi, operand element e (0..7).
// Chosen so that an accumulator (16 fp32 per lane) converts to the next stage's operand
// with no data movement: operand dword d of K-step ms = pack(acc[8ms+2d], acc[8ms+2d+1]).
FFC_HD constexpr int kslot_row(int ms, int hi, int e) { return 16 * ms + 8 * (e >> 2) + 4 * hi + (e & 3); }

enum { DT_BF16 = 0, DT_F16 = 1 };

template <int N1_, int N2_, int N3_>
struct Geo {
  static constexpr int N1 = N1_, N2 = N2_, N3 = N3_;
  static constexpr int N = N1 * N2 * N3;
  static constexpr int Mi = N2 * N3;            // inner (in-register) transform length
  static constexpr bool OUTER = (N1 > 1);
  static constexpr int S1 = OUTER ? 32 / N1 : 1;  // column sets per outer tile (block-diag)
  static constexpr int SU = 32 / N2, SV = 32 / N3;
  static constexpr int G = SU * SV;              // E rows (k1 values / sequences) per inner tile
  static constexpr int ROWS = OUTER ? N1 : G;    // E rows per unit
  static constexpr int NT = ROWS / G;            // inner tiles per unit
  static constexpr int NW = OUTER ? N / 4096 : 1;  // waves cooperating on one unit
  static constexpr int TPW = NT / NW;            // inner tiles per wave (4 when OUTER)
  static constexpr int ECPLX = ROWS * Mi;        // complex points held in LDS per unit
  static constexpr int PLANE = ECPLX * 2;        // bytes per (re|im) plane
  static constexpr int EBYTES = PLANE * 2;
  static constexpr int CR = N3 / 4;              // 8-byte chunks per n2-row
  static constexpr int PER = 64 / N3;            // n2 period of the bank swizzle
  // Workgroup = 8 waves; it works on UPW independent units at a time (lock-step).
  static constexpr int WGW = 8;
  static constexpr int UPW = WGW / NW;
  // LDS map: [UPW exchange buffers][tables copied from the plan blob at kernel start]
  static constexpr int L_E = 0;
  static constexpr int L_F1 = UPW * EBYTES;
  static constexpr int L_F2 = L_F1 + (OUTER ? 6144 : 0);
  static constexpr int L_F3 = L_F2 + 6144;
  static constexpr int L_TW = L_F3 + (N3 != N2 ? 6144 : 0);
  static constexpr int L_TW2 = L_TW + 8192;
  static constexpr bool TW2_SEP = (N3 != N2) || !OUTER;   // separate inverse inner-twiddle table
  static constexpr int L_BASE = L_TW2 + (TW2_SEP ? 8192 : 0);
  static constexpr bool HAS_SP = OUTER && N2 == 32 && N3 == 32;    // frequency-sparse kernel variant (PlanTabs::mat_sp)
  static constexpr int L_FS = L_BASE;
  static constexpr int LDS_BYTES = L_BASE + (HAS_SP ? 3072 : 0);   // outer twiddles are generated on the fly (no tables)
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(N2 == 16 || N2 == 32, "");
  static_assert(N3 == 16 || N3 == 32, "");
  static_assert(N1 == 1 || N1 == 16 || N1 == 32, "");
  static_assert(!OUTER || TPW == 4, "");
};

// Byte offset inside one plane of element (row, m); chunks of 4 elements (8 B) are XOR-swizzled
// inside their n2-row so that lane<->n2 strided b64 stores and tr_b16 reads are bank-conflict free.
// I is int on the device/host and a 64-lane int vector in the CPU wave simulator.
template <class GEO, class I>
FFC_HD inline __attribute__((always_inline)) I e_off(I row, I m) {
  I n2 = m / GEO::N3, n3 = m % GEO::N3;
  I cw = n3 >> 2;
  I sig = (n2 / GEO::PER) % GEO::CR;
  return row * (GEO::Mi * 2) + (n2 * GEO::CR + (cw ^ sig)) * 8 + (m & 3) * 2;
}

// Internal ("Monarch order") k_f layout: per h, NT tiles of 1024 complex; position
//   ((tau*8 + rho)*32 + U)*4 + v   with V = 4*rho + v = sV*N3 + k3, U = sU*N2 + k2.
// Returns the natural frequency index f held at that position.
template <class GEO>
FFC_HD constexpr int kf_freq(int tau, int V, int U) {
  int sV = V / GEO::N3, k3 = V % GEO::N3, sU = U / GEO::N2, k2 = U % GEO::N2;
  int k1 = GEO::OUTER ? tau * GEO::G + sU * GEO::SV + sV : 0;
  return k1 + GEO::N1 * (k2 + GEO::N2 * k3);
}

}  // namespace ffc
