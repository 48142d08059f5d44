// Short depthwise conv1d (K odd, usually 3/5/7), HBM-bound streaming kernels for gfx950.
//   y[b,d,l] = bias[d] + sum_k w[d,k] * u[b,d,l+k-P],  L_out = L + 2P - K + 1
// Replaces reference csrc/flashfftconv/conv1d/{conv1d_bhl,conv1d_blh,conv1d_bwd_cuda_bhl,
// conv1d_bwd_cuda_blh}.cu.  Differences by design: fp32 accumulation (the reference accumulates in
// the input dtype, conv1d_bhl.cu:13-43), 16-byte vector accesses along the contiguous axis, and a
// backward that reduces dw/dbias in registers + one fp32 atomic per block instead of materialising
// the (B,D,K,L) im2col tensor (conv1d_bwd_cuda_bhl.cu:10-105).
// Included once per input dtype (FFC_C1D_TI = 0 bf16, 1 fp16, 2 fp32) by ffc_conv1d_t{0,1,2}.hip so that the
// three thirds of the instantiations compile in parallel; ffc_conv1d.hip holds the C-ABI dispatcher.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <type_traits>

#include "../../include/flashfftconv_hip.h"

namespace {

enum { T_BF16 = 0, T_F16 = 1, T_F32 = 2 };

template <int T> struct El;
template <> struct El<T_BF16> {
  using S = uint16_t;
  static __device__ __forceinline__ float ld(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
  static __device__ __forceinline__ uint16_t st(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
};
template <> struct El<T_F16> {
  using S = uint16_t;
  static __device__ __forceinline__ float ld(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
  static __device__ __forceinline__ uint16_t st(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
};
template <> struct El<T_F32> {
  using S = float;
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float f) { return f; }
};

constexpr int MAXK = 15;
constexpr int V = 8;   // elements per thread along the contiguous axis

// aligned vector load of V elements -> float; `ok` false -> zeros
template <int T>
__device__ __forceinline__ void vload(const typename El<T>::S* p, bool ok, float (&v)[V]) {
  if (!ok) {
#pragma unroll
    for (int i = 0; i < V; i++) v[i] = 0.f;
    return;
  }
  if constexpr (T == T_F32) {
    float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    uint4 a = *(const uint4*)p;
    uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = El<T>::ld((uint16_t)(w[i] & 0xffff)); v[2 * i + 1] = El<T>::ld((uint16_t)(w[i] >> 16)); }
  }
}
template <int T>
__device__ __forceinline__ void vstore(typename El<T>::S* p, const float (&v)[V]) {
  if constexpr (T == T_F32) {
    ((float4*)p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    ((float4*)p)[1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (uint32_t)El<T>::st(v[2 * i]) | ((uint32_t)El<T>::st(v[2 * i + 1]) << 16);
    *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ---------------------------------------------------------------- BHL forward / input-gradient
// One thread = V consecutive outputs of one (b,d) row.  FLIP: correlate with the flipped kernel
// (du[l] = sum_k w[k] * dout[l + P - k]).  Fast path needs Lin % V == 0, Lout % V == 0.
template <int TI, int TW, bool FLIP, bool FAST>
__global__ void bhl_kernel(const typename El<TI>::S* __restrict__ u, const typename El<TW>::S* __restrict__ w,
                           const typename El<TW>::S* __restrict__ bias, typename El<TI>::S* __restrict__ y, int D, int Lin,
                           int Lout, int K, int P) {
  const int nch = (Lout + V * (int)blockDim.x - 1) / (V * (int)blockDim.x);
  const int row = blockIdx.x / nch;   // b*D + d
  const int d = row % D;
  const int l0 = ((blockIdx.x % nch) * blockDim.x + threadIdx.x) * V;
  if (l0 >= Lout) return;
  float wk[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; k++) wk[k] = k < K ? El<TW>::ld(w[(size_t)d * K + (FLIP ? K - 1 - k : k)]) : 0.f;
  const float b0 = bias ? El<TW>::ld(bias[d]) : 0.f;
  const typename El<TI>::S* ur = u + (size_t)row * Lin;
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; i++) acc[i] = b0;
  const int shift = FLIP ? (K - 1 - P) : P;   // input index = l + k - shift
  if (FAST) {
    float x[3 * V];
    float t[V];
    vload<TI>(ur + l0 - V, l0 - V >= 0, t);
#pragma unroll
    for (int i = 0; i < V; i++) x[i] = t[i];
    vload<TI>(ur + l0, l0 + V <= Lin, t);
#pragma unroll
    for (int i = 0; i < V; i++) x[V + i] = t[i];
    vload<TI>(ur + l0 + V, l0 + 2 * V <= Lin, t);
#pragma unroll
    for (int i = 0; i < V; i++) x[2 * V + i] = t[i];
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
      if (k < K) {
#pragma unroll
        for (int i = 0; i < V; i++) {
          int idx = V + i + k - shift;   // within [V - shift, 2V + K - 2 - shift] subset of [0, 3V)
          acc[i] += wk[k] * x[idx];
        }
      }
    }
    vstore<TI>(y + (size_t)row * Lout + l0, acc);
  } else {
    for (int i = 0; i < V; i++) {
      int l = l0 + i;
      if (l >= Lout) break;
      float a = b0;
      for (int k = 0; k < K; k++) {
        int j = l + k - shift;
        if (j >= 0 && j < Lin) a += wk[k] * El<TI>::ld(ur[j]);
      }
      y[(size_t)row * Lout + l] = El<TI>::st(a);
    }
  }
}

// dw[d,k] += sum_{l} dout[b,d,l] * u[b,d,l+k-P];  dbias[d] += sum_l dout[b,d,l]   (one (b,d) row chunk per block)
template <int TI>
__global__ void bhl_wgrad_kernel(const typename El<TI>::S* __restrict__ dout, const typename El<TI>::S* __restrict__ u,
                                 float* __restrict__ dw, float* __restrict__ dbias, int D, int L, int Lout, int K, int P, int nch) {
  const int row = blockIdx.x / nch;
  const int d = row % D;
  const typename El<TI>::S* dr = dout + (size_t)row * Lout;
  const typename El<TI>::S* ur = u + (size_t)row * L;
  float acc[MAXK + 1];
#pragma unroll
  for (int k = 0; k <= MAXK; k++) acc[k] = 0.f;
  for (int l = (blockIdx.x % nch) * blockDim.x + threadIdx.x; l < Lout; l += nch * blockDim.x) {
    float g = El<TI>::ld(dr[l]);
    acc[MAXK] += g;
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
      if (k < K) {
        int j = l + k - P;
        if (j >= 0 && j < L) acc[k] += g * El<TI>::ld(ur[j]);
      }
    }
  }
  __shared__ float red[MAXK + 1][4];
#pragma unroll
  for (int k = 0; k <= MAXK; k++) {
    float v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x <= MAXK) {
    int k = threadIdx.x;
    float v = 0.f;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); wv++) v += red[k][wv];
    if (k == MAXK) atomicAdd(&dbias[d], v);
    else if (k < K) atomicAdd(&dw[(size_t)d * K + k], v);
  }
}

// BHL forward for "same" padding with a compile-time K: static tap indices (no dynamic register indexing),
// one 2048-element chunk of a (b,d) row per block.
template <int TI, int TW, int K>
__global__ __launch_bounds__(256) void bhl_same_kernel(const typename El<TI>::S* __restrict__ u, const typename El<TW>::S* __restrict__ w,
                                                       const typename El<TW>::S* __restrict__ bias, typename El<TI>::S* __restrict__ y,
                                                       int D, int L, int nch) {
  constexpr int P = (K - 1) / 2;
  const int row = blockIdx.x / nch;
  const int d = row % D;
  const int l0 = ((blockIdx.x % nch) * 256 + threadIdx.x) * V;
  if (l0 >= L) return;
  float wk[K];
#pragma unroll
  for (int k = 0; k < K; k++) wk[k] = El<TW>::ld(w[(size_t)d * K + k]);
  const float b0 = bias ? El<TW>::ld(bias[d]) : 0.f;
  const typename El<TI>::S* ur = u + (size_t)row * L;
  float x[3 * V], t[V];
  vload<TI>(ur + l0 - V, l0 - V >= 0, t);
#pragma unroll
  for (int i = 0; i < V; i++) x[i] = t[i];
  vload<TI>(ur + l0, true, t);
#pragma unroll
  for (int i = 0; i < V; i++) x[V + i] = t[i];
  vload<TI>(ur + l0 + V, l0 + 2 * V <= L, t);
#pragma unroll
  for (int i = 0; i < V; i++) x[2 * V + i] = t[i];
  float o[V];
#pragma unroll
  for (int i = 0; i < V; i++) {
    float s = b0;
#pragma unroll
    for (int k = 0; k < K; k++) s += wk[k] * x[V + i + k - P];
    o[i] = s;
  }
  vstore<TI>(y + (size_t)row * L + l0, o);
}

// Raw (unconverted) V-element vector: lets the next iteration's loads be issued before the current one is consumed.
template <int T> struct Raw {
  uint4 a;
  __device__ __forceinline__ void load(const typename El<T>::S* p) { a = *(const uint4*)p; }
  __device__ __forceinline__ void get(float (&v)[V], bool ok) const {
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[2 * i] = ok ? El<T>::ld((uint16_t)(w[i] & 0xffff)) : 0.f;
      v[2 * i + 1] = ok ? El<T>::ld((uint16_t)(w[i] >> 16)) : 0.f;
    }
  }
};
template <> struct Raw<T_F32> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = ((const float4*)p)[0]; b = ((const float4*)p)[1]; }
  __device__ __forceinline__ void get(float (&v)[V], bool ok) const {
    const float t[V] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < V; i++) v[i] = ok ? t[i] : 0.f;
  }
};

// Fused BHL backward for "same" padding (Lout == L, P == (K-1)/2), K a compile-time constant: one pass over
// dout and u (16-byte loads) produces du (16-byte stores) and the dw / dbias partial sums.
// A block owns channel d of NB consecutive batches (rows (b0+r)*D + d): the K+1 sums stay in registers across
// all NB rows and leave through one reduction + K+1 atomics per block (one block per row paid that tail, a
// barrier and 4 atomics, for every 4 loop iterations).  The walk over (row, vector) is one flat loop whose next
// iteration's six loads are issued before the current vectors are consumed; neighbour vectors are loaded
// unconditionally from clamped addresses and masked (no exec-mask branch per load).
template <int TI, int TW, int K>
__global__ __launch_bounds__(256) void bhl_bwd_kernel(const typename El<TI>::S* __restrict__ dout,
                                                      const typename El<TI>::S* __restrict__ u,
                                                      const typename El<TW>::S* __restrict__ w, typename El<TI>::S* __restrict__ du,
                                                      float* __restrict__ dw, float* __restrict__ dbias, int B, int D, int L, int NB) {
  constexpr int P = (K - 1) / 2;
  static_assert(K - 1 <= V, "neighbour vectors cover K-1 <= V taps");
  const int d = blockIdx.x % D;
  const int b0 = (blockIdx.x / D) * NB;
  const int nrows = min(NB, B - b0);
  const int nv = L / V;                       // vectors per row
  const int total = nrows * nv;
  float wk[K];
#pragma unroll
  for (int k = 0; k < K; k++) wk[k] = El<TW>::ld(w[(size_t)d * K + k]);
  float acc[K + 1];
#pragma unroll
  for (int k = 0; k <= K; k++) acc[k] = 0.f;

  Raw<TI> gq[3], xq[3];
  auto issue = [&](int id) {
    id = min(id, total - 1);
    const int r = id / nv, c = id - r * nv;
    const size_t base = ((size_t)(b0 + r) * D + d) * L;
    const int lm = max(c - 1, 0) * V, l0 = c * V, lp = min(c + 1, nv - 1) * V;
    gq[0].load(dout + base + lm); gq[1].load(dout + base + l0); gq[2].load(dout + base + lp);
    xq[0].load(u + base + lm); xq[1].load(u + base + l0); xq[2].load(u + base + lp);
  };
  if ((int)threadIdx.x < total) issue(threadIdx.x);
  for (int id = threadIdx.x; id < total; id += 256) {
    const int r = id / nv, c = id - r * nv;
    float g[3 * V], x[3 * V], t[V];
    gq[0].get(t, c > 0);
#pragma unroll
    for (int i = 0; i < V; i++) g[i] = t[i];
    gq[1].get(t, true);
#pragma unroll
    for (int i = 0; i < V; i++) g[V + i] = t[i];
    gq[2].get(t, c + 1 < nv);
#pragma unroll
    for (int i = 0; i < V; i++) g[2 * V + i] = t[i];
    xq[0].get(t, c > 0);
#pragma unroll
    for (int i = 0; i < V; i++) x[i] = t[i];
    xq[1].get(t, true);
#pragma unroll
    for (int i = 0; i < V; i++) x[V + i] = t[i];
    xq[2].get(t, c + 1 < nv);
#pragma unroll
    for (int i = 0; i < V; i++) x[2 * V + i] = t[i];
    if (id + 256 < total) issue(id + 256);     // next iteration's loads in flight behind this one's math + store
    float o[V];
#pragma unroll
    for (int i = 0; i < V; i++) {
      // du[l] = sum_k w[k] * dout[l + P - k];  dw[k] += dout[l] * u[l + k - P]
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < K; k++) {
        s += wk[k] * g[V + i + P - k];
        acc[k] += g[V + i] * x[V + i + k - P];
      }
      o[i] = s;
      acc[K] += g[V + i];
    }
    vstore<TI>(du + ((size_t)(b0 + r) * D + d) * L + c * V, o);
  }
  __shared__ float red[K + 1][4];
#pragma unroll
  for (int k = 0; k <= K; k++) {
    float v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x <= K) {
    const int k = threadIdx.x;
    float v = red[k][0] + red[k][1] + red[k][2] + red[k][3];
    if (k == K) atomicAdd(&dbias[d], v);
    else atomicAdd(&dw[(size_t)d * K + k], v);
  }
}

// ---------------------------------------------------------------- BLH forward / input-gradient
// u (B,L,D), w (K,D).  One thread = V consecutive channels x TL consecutive positions.
constexpr int TL = 4;
template <int TI, int TW, bool FLIP, bool FAST>
__global__ void blh_kernel(const typename El<TI>::S* __restrict__ u, const typename El<TW>::S* __restrict__ w,
                           const typename El<TW>::S* __restrict__ bias, typename El<TI>::S* __restrict__ y, int D, int Lin,
                           int Lout, int K, int P) {
  const int d0 = (blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (d0 >= D) return;
  const int b = blockIdx.z;
  const int l0 = blockIdx.y * TL;
  const int shift = FLIP ? (K - 1 - P) : P;
  float acc[TL][V];
  float bv[V];
  if (bias) {
    if (FAST) vload<TW>(bias + d0, true, bv);
    else
      for (int i = 0; i < V; i++) bv[i] = d0 + i < D ? El<TW>::ld(bias[d0 + i]) : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < V; i++) bv[i] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < TL; t++)
#pragma unroll
    for (int i = 0; i < V; i++) acc[t][i] = bv[i];
  const typename El<TI>::S* ub = u + (size_t)b * Lin * D;
  for (int r = 0; r < TL + K - 1; r++) {   // input row j feeds outputs l = j - k + shift
    int j = l0 - shift + r;
    if (j < 0 || j >= Lin) continue;
    float x[V];
    if (FAST) vload<TI>(ub + (size_t)j * D + d0, true, x);
    else
      for (int i = 0; i < V; i++) x[i] = d0 + i < D ? El<TI>::ld(ub[(size_t)j * D + d0 + i]) : 0.f;
#pragma unroll
    for (int t = 0; t < TL; t++) {
      int k = r - t;   // l = l0 + t = j - k + shift
      if (k < 0 || k >= K) continue;
      int kk = FLIP ? K - 1 - k : k;
      float wv[V];
      if (FAST) vload<TW>(w + (size_t)kk * D + d0, true, wv);
      else
        for (int i = 0; i < V; i++) wv[i] = d0 + i < D ? El<TW>::ld(w[(size_t)kk * D + d0 + i]) : 0.f;
#pragma unroll
      for (int i = 0; i < V; i++) acc[t][i] += wv[i] * x[i];
    }
  }
#pragma unroll
  for (int t = 0; t < TL; t++) {
    int l = l0 + t;
    if (l >= Lout) break;
    typename El<TI>::S* yp = y + ((size_t)b * Lout + l) * D + d0;
    if (FAST) vstore<TI>(yp, acc[t]);
    else
      for (int i = 0; i < V; i++)
        if (d0 + i < D) yp[i] = El<TI>::st(acc[t][i]);
  }
}

// dw[k,d], dbias[d] for BLH: thread = V channels, block loops over a slab of (b,l) rows.
// (launched with 64 threads: without the bound the compiler budgets for 1024-thread blocks = 128 registers and spilled 66 - 77 of this kernel's
// 16 x 8 fp32 sums into scratch memory, hipcc -Rpass-analysis=kernel-resource-usage; round 6)
template <int TI, bool FAST>
__global__ __launch_bounds__(64) void blh_wgrad_kernel(const typename El<TI>::S* __restrict__ dout, const typename El<TI>::S* __restrict__ u,
                                 float* __restrict__ dw, float* __restrict__ dbias, int B, int D, int L, int Lout, int K, int P,
                                 int rows_per_block) {
  const int d0 = (blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (d0 >= D) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t total = (int64_t)B * Lout;
  float acc[MAXK + 1][V];
#pragma unroll
  for (int k = 0; k <= MAXK; k++)
#pragma unroll
    for (int i = 0; i < V; i++) acc[k][i] = 0.f;
  for (int64_t r = r0; r < r0 + rows_per_block && r < total; r++) {
    int b = (int)(r / Lout), l = (int)(r % Lout);
    float g[V];
    if (FAST) vload<TI>(dout + (size_t)r * D + d0, true, g);
    else
      for (int i = 0; i < V; i++) g[i] = d0 + i < D ? El<TI>::ld(dout[(size_t)r * D + d0 + i]) : 0.f;
#pragma unroll
    for (int i = 0; i < V; i++) acc[MAXK][i] += g[i];
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
      if (k < K) {
        int j = l + k - P;
        if (j >= 0 && j < L) {
          float x[V];
          const typename El<TI>::S* up = u + ((size_t)b * L + j) * D + d0;
          if (FAST) vload<TI>(up, true, x);
          else
            for (int i = 0; i < V; i++) x[i] = d0 + i < D ? El<TI>::ld(up[i]) : 0.f;
#pragma unroll
          for (int i = 0; i < V; i++) acc[k][i] += g[i] * x[i];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; i++) {
    if (d0 + i < D) {
      atomicAdd(&dbias[d0 + i], acc[MAXK][i]);
#pragma unroll
      for (int k = 0; k < MAXK; k++)
        if (k < K) atomicAdd(&dw[(size_t)k * D + d0 + i], acc[k][i]);
    }
  }
}

thread_local std::string g_err1d;
int fail1d(const char* m);

template <int TI, int TW>
int launch_fwd(const void* u, const void* w, const void* bias, void* y, int64_t B, int64_t D, int64_t Lin, int64_t Lout, int K,
               int P, bool bhl, bool flip, hipStream_t st) {
  using SI = typename El<TI>::S;
  using SW = typename El<TW>::S;
  const int esz = sizeof(SI);
  if (bhl) {
    // the fast path keeps inputs [l0 - V, l0 + 2V) in registers: tap index V + i + k - shift must stay inside [0, 3V)
    const int shift = flip ? (K - 1 - P) : P;
    bool fast = (Lin % V == 0) && (Lout % V == 0) && (K - 1 <= V) && shift >= 0 && shift <= V && (K - 1 - shift) <= V &&
                !(((uintptr_t)u | (uintptr_t)y) & 15) && esz * V % 16 == 0;
    int64_t nblk = ((Lout + V * 256 - 1) / (V * 256)) * B * D;
    if (nblk > 2147483647LL) return fail1d("grid too large");
    dim3 block(256), grid((unsigned)nblk);
#define FFC_L(FL, FA) hipLaunchKernelGGL((bhl_kernel<TI, TW, FL, FA>), grid, block, 0, st, (const SI*)u, (const SW*)w, (const SW*)bias, (SI*)y, (int)D, (int)Lin, (int)Lout, K, P)
    if (flip) { if (fast) FFC_L(true, true); else FFC_L(true, false); }
    else { if (fast) FFC_L(false, true); else FFC_L(false, false); }
#undef FFC_L
  } else {
    bool fast = (D % V == 0) && !(((uintptr_t)u | (uintptr_t)y | (uintptr_t)w | (uintptr_t)bias) & 15);
    int tx = 64;
    dim3 block(tx), grid((unsigned)((D + V * tx - 1) / (V * tx)), (unsigned)((Lout + TL - 1) / TL), (unsigned)B);
#define FFC_L(FL, FA) hipLaunchKernelGGL((blh_kernel<TI, TW, FL, FA>), grid, block, 0, st, (const SI*)u, (const SW*)w, (const SW*)bias, (SI*)y, (int)D, (int)Lin, (int)Lout, K, P)
    if (flip) { if (fast) FFC_L(true, true); else FFC_L(true, false); }
    else { if (fast) FFC_L(false, true); else FFC_L(false, false); }
#undef FFC_L
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail1d(hipGetErrorString(e));
  return 0;
}

// Fused BLH backward for "same" padding, K in {3, 5, 7}: a thread owns V channels and walks a slab of consecutive
// positions of one batch with K-row sliding windows of dout and u in registers (slot = row mod K, static after
// unrolling by K): every row of dout / u is loaded once (16 bytes per lane, 1 KiB per wave), du is stored as it
// goes, dw / dbias sums leave through one fp32 atomic per value at the end of the slab.
template <int TI, int TW, int K>
__global__ __launch_bounds__(64) void blh_bwd_kernel(const typename El<TI>::S* __restrict__ dout, const typename El<TI>::S* __restrict__ u,
                                                     const typename El<TW>::S* __restrict__ w, typename El<TI>::S* __restrict__ du,
                                                     float* __restrict__ dw, float* __restrict__ dbias, int D, int L, int R) {
  constexpr int P = (K - 1) / 2;
  const int d0 = (blockIdx.x * 64 + threadIdx.x) * V;
  if (d0 >= D) return;
  const int slabs = (L + R - 1) / R;
  const int b = blockIdx.y / slabs;
  const int r0 = (blockIdx.y % slabs) * R;
  const int r1 = min(r0 + R, L);
  const size_t base = (size_t)b * L * D + d0;
  float wk[K][V], g[K][V], x[K][V], acc[K + 1][V];
#pragma unroll
  for (int k = 0; k < K; k++) {
    vload<TW>(w + (size_t)k * D + d0, true, wk[k]);
#pragma unroll
    for (int i = 0; i < V; i++) { acc[k][i] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < V; i++) acc[K][i] = 0.f;
  // rows r0-P .. r0+P-1 -> slots (j - r0 + P) mod K
#pragma unroll
  for (int t = 0; t < 2 * P; t++) {
    const int j = r0 - P + t;
    vload<TI>(dout + base + (size_t)j * D, j >= 0 && j < L, g[t % K]);
    vload<TI>(u + base + (size_t)j * D, j >= 0 && j < L, x[t % K]);
  }
  for (int lb = r0; lb < r1; lb += K) {
#pragma unroll
    for (int s = 0; s < K; s++) {
      const int l = lb + s;
      if (l < r1) {
        const int jn = l + P;
        vload<TI>(dout + base + (size_t)jn * D, jn < L, g[(s + K - 1) % K]);
        vload<TI>(u + base + (size_t)jn * D, jn < L, x[(s + K - 1) % K]);
        float o[V];
#pragma unroll
        for (int i = 0; i < V; i++) {
          float sum = 0.f;
          const float gl = g[(s + P) % K][i];
#pragma unroll
          for (int k = 0; k < K; k++) {
            sum += wk[k][i] * g[(s + 2 * P - k) % K][i];     // du[l] = sum_k w[k] dout[l + P - k]
            acc[k][i] += gl * x[(s + k) % K][i];              // dw[k] += dout[l] u[l + k - P]
          }
          acc[K][i] += gl;
          o[i] = sum;
        }
        vstore<TI>(du + base + (size_t)l * D, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; i++) {
    atomicAdd(&dbias[d0 + i], acc[K][i]);
#pragma unroll
    for (int k = 0; k < K; k++) atomicAdd(&dw[(size_t)k * D + d0 + i], acc[k][i]);
  }
}

// BLH forward for "same" padding, K in {3, 5, 7}: same sliding-window walk as blh_bwd_kernel, every input row is
// loaded once (the tiled kernel above re-reads (TL+K-1)/TL rows per output row).
template <int TI, int TW, int K>
__global__ __launch_bounds__(64) void blh_same_kernel(const typename El<TI>::S* __restrict__ u, const typename El<TW>::S* __restrict__ w,
                                                      const typename El<TW>::S* __restrict__ bias, typename El<TI>::S* __restrict__ y,
                                                      int D, int L, int R) {
  constexpr int P = (K - 1) / 2;
  const int d0 = (blockIdx.x * 64 + threadIdx.x) * V;
  if (d0 >= D) return;
  const int slabs = (L + R - 1) / R;
  const int b = blockIdx.y / slabs;
  const int r0 = (blockIdx.y % slabs) * R;
  const int r1 = min(r0 + R, L);
  const size_t base = (size_t)b * L * D + d0;
  float wk[K][V], x[K][V], bv[V];
#pragma unroll
  for (int k = 0; k < K; k++) vload<TW>(w + (size_t)k * D + d0, true, wk[k]);
  vload<TW>(bias + d0, bias != nullptr, bv);
#pragma unroll
  for (int t = 0; t < 2 * P; t++) {
    const int j = r0 - P + t;
    vload<TI>(u + base + (size_t)j * D, j >= 0 && j < L, x[t % K]);
  }
  for (int lb = r0; lb < r1; lb += K) {
#pragma unroll
    for (int s = 0; s < K; s++) {
      const int l = lb + s;
      if (l < r1) {
        const int jn = l + P;
        vload<TI>(u + base + (size_t)jn * D, jn < L, x[(s + K - 1) % K]);
        float o[V];
#pragma unroll
        for (int i = 0; i < V; i++) {
          float sum = bv[i];
#pragma unroll
          for (int k = 0; k < K; k++) sum += wk[k][i] * x[(s + k) % K][i];     // y[l] = sum_k w[k] u[l + k - P]
          o[i] = sum;
        }
        vstore<TI>(y + base + (size_t)l * D, o);
      }
    }
  }
}

template <int TI>
int launch_wgrad(const void* dout, const void* u, float* dw, float* dbias, int64_t B, int64_t D, int64_t L, int64_t Lout, int K,
                 int P, bool bhl, hipStream_t st) {
  using SI = typename El<TI>::S;
  if (bhl) {
    int nch = (int)std::min<int64_t>((Lout + 255) / 256, 8);
    int64_t nblk = (int64_t)nch * B * D;
    if (nblk > 2147483647LL) return fail1d("grid too large");
    dim3 block(256), grid((unsigned)nblk);
    hipLaunchKernelGGL((bhl_wgrad_kernel<TI>), grid, block, 0, st, (const SI*)dout, (const SI*)u, dw, dbias, (int)D, (int)L,
                       (int)Lout, K, P, nch);
  } else {
    bool fast = (D % V == 0) && !(((uintptr_t)u | (uintptr_t)dout) & 15);
    int tx = 64;
    int64_t total = B * Lout;
    int rpb = (int)std::max<int64_t>(64, (total + 1023) / 1024);
    dim3 block(tx), grid((unsigned)((D + V * tx - 1) / (V * tx)), (unsigned)((total + rpb - 1) / rpb));
    if (fast)
      hipLaunchKernelGGL((blh_wgrad_kernel<TI, true>), grid, block, 0, st, (const SI*)dout, (const SI*)u, dw, dbias, (int)B, (int)D,
                         (int)L, (int)Lout, K, P, rpb);
    else
      hipLaunchKernelGGL((blh_wgrad_kernel<TI, false>), grid, block, 0, st, (const SI*)dout, (const SI*)u, dw, dbias, (int)B,
                         (int)D, (int)L, (int)Lout, K, P, rpb);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail1d(hipGetErrorString(e));
  return 0;
}

template <class F>
int by_dtypes(int ti, int tw, F f) {
#define FFC_C(a, b) if constexpr (a == FFC_C1D_TI) { if (ti == a && tw == b) return f(std::integral_constant<int, a>(), std::integral_constant<int, b>()); }
  FFC_C(0, 0) FFC_C(0, 1) FFC_C(0, 2) FFC_C(1, 0) FFC_C(1, 1) FFC_C(1, 2) FFC_C(2, 0) FFC_C(2, 1) FFC_C(2, 2)
#undef FFC_C
  return fail1d("unsupported dtype combination");
}

}  // namespace

// error plumbing shared with ffc_hip.hip
extern "C" void ffc_set_error_(const char* m);
namespace { int fail1d(const char* m) { ffc_set_error_(m); return 1; } }

#define FFC_C1D_CAT2(a, b) a##b
#define FFC_C1D_CAT(a, b) FFC_C1D_CAT2(a, b)
#define FFC_C1D_NAME(p) FFC_C1D_CAT(p, FFC_C1D_TI)
extern "C" {

int FFC_C1D_NAME(ffc_c1d_fwd_)(const void* u, const void* w, const void* bias, void* y, int in_dtype, int w_dtype, int64_t B, int64_t D,
                   int64_t L, int K, int P, int is_bhl, void* stream) {
  if (!u || !w || !y) return fail1d("null arg");
  if (K < 1 || K > MAXK || (K % 2) != 1) return fail1d("kernel size must be odd and <= 15");
  int64_t Lout = L + 2 * (int64_t)P - K + 1;
  if (Lout <= 0) return fail1d("empty output");
  if (is_bhl && Lout == L && 2 * P == K - 1 && K <= 9 && L % V == 0 && !(((uintptr_t)u | (uintptr_t)y) & 15)) {
    const int nch = (int)((L + V * 256 - 1) / (V * 256));
    if ((int64_t)nch * B * D <= 2147483647LL)
      return by_dtypes(in_dtype, w_dtype, [&](auto ti, auto tw) {
        constexpr int TI = decltype(ti)::value, TW = decltype(tw)::value;
        using SI = typename El<TI>::S;
        using SW = typename El<TW>::S;
        dim3 block(256), grid((unsigned)(nch * B * D));
#define FFC_L(KK) hipLaunchKernelGGL((bhl_same_kernel<TI, TW, KK>), grid, block, 0, (hipStream_t)stream, (const SI*)u, (const SW*)w, (const SW*)bias, (SI*)y, (int)D, (int)L, nch)
        switch (K) { case 1: FFC_L(1); break; case 3: FFC_L(3); break; case 5: FFC_L(5); break; case 7: FFC_L(7); break; default: FFC_L(9); break; }
#undef FFC_L
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : fail1d(hipGetErrorString(e));
      });
  }
  // (K = 3 stays on the tiled kernel: 1.5 row reads per output row, measured faster than the serial walk)
  if (!is_bhl && Lout == L && 2 * P == K - 1 && (K == 5 || K == 7) && D % V == 0 &&
      !(((uintptr_t)u | (uintptr_t)y | (uintptr_t)w | (uintptr_t)bias) & 15)) {
    int R = (int)std::max<int64_t>(32, (B * L + 2047) / 2048);      // ~2048 slabs
    R = (int)std::min<int64_t>(R, L);
    const int64_t slabs = (L + R - 1) / R;
    if (B * slabs <= 65535)
      return by_dtypes(in_dtype, w_dtype, [&](auto ti, auto tw) {
        constexpr int TI = decltype(ti)::value, TW = decltype(tw)::value;
        using SI = typename El<TI>::S;
        using SW = typename El<TW>::S;
        dim3 block(64), grid((unsigned)((D / V + 63) / 64), (unsigned)(B * slabs));
#define FFC_L(KK) hipLaunchKernelGGL((blh_same_kernel<TI, TW, KK>), grid, block, 0, (hipStream_t)stream, (const SI*)u, (const SW*)w, (const SW*)bias, (SI*)y, (int)D, (int)L, R)
        if (K == 5) FFC_L(5); else FFC_L(7);
#undef FFC_L
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : fail1d(hipGetErrorString(e));
      });
  }
  return by_dtypes(in_dtype, w_dtype, [&](auto ti, auto tw) {
    return launch_fwd<decltype(ti)::value, decltype(tw)::value>(u, w, bias, y, B, D, L, Lout, K, P, is_bhl != 0, false, (hipStream_t)stream);
  });
}

int FFC_C1D_NAME(ffc_c1d_bwd_)(const void* dout, const void* u, const void* w, void* du, float* dw, float* dbias, int in_dtype, int w_dtype,
                   int64_t B, int64_t D, int64_t L, int K, int P, int is_bhl, void* stream) {
  if (!dout || !u || !w || !du || !dw || !dbias) return fail1d("null arg");
  if (K < 1 || K > MAXK || (K % 2) != 1) return fail1d("kernel size must be odd and <= 15");
  int64_t Lout = L + 2 * (int64_t)P - K + 1;
  if (Lout <= 0) return fail1d("empty output");
  // BHL, "same" padding, aligned rows: one fused pass
  if (is_bhl && Lout == L && 2 * P == K - 1 && K <= 9 && L % V == 0 && B * D <= 2147483647LL &&
      !(((uintptr_t)u | (uintptr_t)dout | (uintptr_t)du) & 15)) {
    int rc = by_dtypes(in_dtype, w_dtype, [&](auto ti, auto tw) {
      constexpr int TI = decltype(ti)::value, TW = decltype(tw)::value;
      using SI = typename El<TI>::S;
      using SW = typename El<TW>::S;
      // batches per block: keep >= ~8K blocks in the grid, at most 16 rows per block
      const int NB = (int)std::max<int64_t>(1, std::min<int64_t>(16, B * D / 8192));
      dim3 block(256), grid((unsigned)(D * ((B + NB - 1) / NB)));
#define FFC_L(KK) hipLaunchKernelGGL((bhl_bwd_kernel<TI, TW, KK>), grid, block, 0, (hipStream_t)stream, (const SI*)dout, (const SI*)u, (const SW*)w, (SI*)du, dw, dbias, (int)B, (int)D, (int)L, NB)
      switch (K) { case 1: FFC_L(1); break; case 3: FFC_L(3); break; case 5: FFC_L(5); break; case 7: FFC_L(7); break; default: FFC_L(9); break; }
#undef FFC_L
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : fail1d(hipGetErrorString(e));
    });
    return rc;
  }
  // BLH, "same" padding, K in {3,5,7}, aligned channel vectors: one fused pass
  if (!is_bhl && Lout == L && 2 * P == K - 1 && (K == 3 || K == 5 || K == 7) && D % V == 0 &&
      !(((uintptr_t)u | (uintptr_t)dout | (uintptr_t)du | (uintptr_t)w) & 15)) {
    return by_dtypes(in_dtype, w_dtype, [&](auto ti, auto tw) {
      constexpr int TI = decltype(ti)::value, TW = decltype(tw)::value;
      using SI = typename El<TI>::S;
      using SW = typename El<TW>::S;
      int R = (int)std::max<int64_t>(64, (B * L + 1023) / 1024);      // ~1024 slabs
      R = (int)std::min<int64_t>(R, L);
      const int64_t slabs = (L + R - 1) / R;
      if (B * slabs > 65535) return fail1d("grid too large");
      dim3 block(64), grid((unsigned)((D / V + 63) / 64), (unsigned)(B * slabs));
      if (K == 3) hipLaunchKernelGGL((blh_bwd_kernel<TI, TW, 3>), grid, block, 0, (hipStream_t)stream, (const SI*)dout, (const SI*)u, (const SW*)w, (SI*)du, dw, dbias, (int)D, (int)L, R);
      else if (K == 5) hipLaunchKernelGGL((blh_bwd_kernel<TI, TW, 5>), grid, block, 0, (hipStream_t)stream, (const SI*)dout, (const SI*)u, (const SW*)w, (SI*)du, dw, dbias, (int)D, (int)L, R);
      else hipLaunchKernelGGL((blh_bwd_kernel<TI, TW, 7>), grid, block, 0, (hipStream_t)stream, (const SI*)dout, (const SI*)u, (const SW*)w, (SI*)du, dw, dbias, (int)D, (int)L, R);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : fail1d(hipGetErrorString(e));
    });
  }
  // du[l] = sum_k w[k] * dout[l + P - k]: the same streaming kernel with the flipped kernel
  int rc = by_dtypes(in_dtype, w_dtype, [&](auto ti, auto tw) {
    return launch_fwd<decltype(ti)::value, decltype(tw)::value>(dout, w, nullptr, du, B, D, Lout, L, K, P, is_bhl != 0, true, (hipStream_t)stream);
  });
  if (rc) return rc;
  return by_dtypes(in_dtype, 0, [&](auto ti, auto) {
    return launch_wgrad<decltype(ti)::value>(dout, u, dw, dbias, B, D, L, Lout, K, P, is_bhl != 0, (hipStream_t)stream);
  });
}

}  // extern "C"
