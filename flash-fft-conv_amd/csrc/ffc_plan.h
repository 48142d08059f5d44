// Host-side plan: the DFT-tile operands and twiddle tables of one FFT size, precomputed in
// float64 and laid out exactly as the kernel's lanes load them.  Replaces the register_buffer
// tables built in reference flashfftconv/conv.py:72-551 (fft_matrix / compute_twiddle_factors_*).
#pragma once
#include <stdint.h>
#include <vector>

namespace ffc {

struct PlanTabs {      // byte offsets into the plan blob
  int mat[3];          // operand tables of digits N1,N2,N3: [6][64][4] u32
  int twin, twin2;     // inner twiddle, fwd / inverse (ctab16)
  int base;            // ctab16 : s_fwd * W_N^{(s1*128 + 4j)*k1}  (outer fwd twiddle of wave 0, tile 0)
  int delta;           // [2 hi][16 r] complex f32 : W_N^{k1}            (tile -> tile+1 step)
  int omega;           // [NW][2][16] complex f32 : W_N^{128*S1*w*k1}    (wave offset)
  int oi_a;            // [NT][32][SV] complex f32 : W_N^{-n2*N3*k1}
  int oi_b;            // [NT][SU][2][16] complex f32 : W_N^{-n3*k1}
  int total;
};

struct HostPlan {
  int N = 0, N1 = 0, N2 = 0, N3 = 0, dtype = 0;
  int NT = 0, NW = 0, G = 0;
  double s_fwd = 1, s_k = 1, s_inv = 1;   // s_fwd * s_k * s_inv == 1/N
  PlanTabs tabs{};
  std::vector<uint8_t> blob;
  std::vector<int32_t> kf_freq;  // internal position -> natural frequency, NT*1024 entries
};

// Supported sizes: 256,512,1024 (inner only) and 4096,8192,16384,32768 (outer x inner).
bool plan_factors(int N, int* n1, int* n2, int* n3);
bool build_plan(int N, int dtype, HostPlan* out);

uint16_t f32_to_bf16(float f);
uint16_t f32_to_f16(float f);
float bf16_to_f32(uint16_t h);
float f16_to_f32(uint16_t h);

}  // namespace ffc
