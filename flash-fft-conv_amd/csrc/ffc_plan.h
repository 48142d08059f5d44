// Host-side plan: the DFT-tile operands and twiddle tables of one FFT size, precomputed in
// float64 and laid out exactly as the kernel's lanes load them.  Replaces the register_buffer
// tables built in reference flashfftconv/conv.py:72-551 (fft_matrix / compute_twiddle_factors_*).
#pragma once
#include <stdint.h>
#include <vector>

// Outer twiddle folded into per-(stage, tile) inner DFT matrices (ffc_body.h tile_fwd; PlanTabs::fold).  0: off; 1 (default): the forward
// kernels of fft 16384, where two pairs share a tile's matrices (measured -6 .. -8 %, profiles/r05_ab_fold_twiddle.txt); 2: also fft 32768
// (forward + saved-spectra backward; measured +-1 %: the L2 fetches of the matrices cost what the removed instructions save)
#ifndef FFC_FOLD_TW
#define FFC_FOLD_TW 1
#endif

namespace ffc {

struct PlanTabs {      // byte offsets into the plan blob
  int mat[3];          // operand tables of digits N1,N2,N3: [6][64][4] u32
  int twin, twin2;     // inner twiddle, fwd / inverse (ctab16)
  // multi-pass sizes (HostPlan::R > 1): outer-digit operand tables of pass k0, [k0][0] forward, [k0][1] inverse
  // (pass 0 uses mat[0] for both): F[k1][n1] = W_N1^{n1 k1} W_{R N1}^{n1 k0}, see ffc_plan.cpp fill_mat_pass
  int matk[4][2];
  // inner-only multi-pass sizes (fft 2048 = 2 passes of the 32 x 32 kernel): per pass [Fa | Finv | tw | tw2] (28672 bytes)
  int ipass[4];
  // frequency-sparse kernels (32-point inner digits): K-step-0 operand table of the 32-point DFT whose contraction slots hold
  // k3 = 0..3 (lane half 0) and 28..31 (lane half 1) -- the only non-zero spectrum rows of a low-pass k_f (ffc_conv_fwd_sparse)
  int mat_sp;
  // Round 5 (build switch FFC_FOLD_TW): outer twiddle folded into per-tile inner DFT matrices, single-pass fft 32768 plans only
  // (0 = absent).  [which 4][tile k1 32][6 x 64 x 16 bytes], same operand layout as mat[]: which 0 = forward stage a
  // (s_fwd W_32^{n2 k2} W_N^{32 n2 k1}, factor on the contraction index n2), 1 = forward stage b (W_32^{n3 k3} W_N^{n3 k1}),
  // 2 = inverse stage b (W_32^{-n3 k3} W_N^{-n3 k1}, factor on the output index n3), 3 = inverse stage a (s_inv W_32^{-n2 k2}
  // W_N^{-32 n2 k1}): W_N^{m k1} with m = 32 n2 + n3 is the product of a factor on n2 and one on n3, and each of them sits on an index
  // that an inner stage contracts (forward) or produces (inverse) -- no elementwise outer twiddle is left (ffc_body.h tile_fwd / tile_inv)
  int fold;
  int total;
};

struct HostPlan {
  int N = 0, N1 = 0, N2 = 0, N3 = 0, dtype = 0;
  // R > 1: fft size N = R * (N1*N2*N3) run as R passes of the fused N1*N2*N3 kernel over the same rows (pass k0 = the
  // frequencies f = k0 (mod R)): the radix-R step is folded into the outer digit's DFT matrix and twiddle chain
  int R = 1;
  int NT = 0, NW = 0, G = 0;        // NT: inner tiles per unit AND per pass (k_f has R*NT tiles per head)
  double s_fwd = 1, s_k = 1, s_inv = 1;   // s_fwd * s_k * s_inv == 1/N
  PlanTabs tabs{};
  std::vector<uint8_t> blob;
  std::vector<int32_t> kf_freq;  // internal position -> natural frequency, NT*1024 entries
};

// Supported sizes: 256,512,1024 (inner only), 2048 (2 passes of 1024), 4096,8192,16384,32768 (outer x inner),
// 65536,131072 (2 / 4 passes of 32768).
bool plan_factors(int N, int* n1, int* n2, int* n3, int* passes = nullptr);
bool build_plan(int N, int dtype, HostPlan* out);

uint16_t f32_to_bf16(float f);
uint16_t f32_to_f16(float f);
float bf16_to_f32(uint16_t h);
float f16_to_f32(uint16_t h);

}  // namespace ffc
