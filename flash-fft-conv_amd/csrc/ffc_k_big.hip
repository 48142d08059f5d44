// HBM-level outer DFT passes for FFT sizes >= 65536 (ffc_big.h) + C-ABI.
#include "ffc_dev.h"
using namespace ffc;

template <int N0, int DT, bool FWD>
__global__ __launch_bounds__(GeoBig<N0>::WGW * 64, 2) void big_kernel(BigArgs a) {
  BigBody<DevB, N0, DT>::template run<FWD>(a, blockIdx.x);
}

template <int N0, int DT, bool FWD>
static int launch_big(const BigArgs& a, hipStream_t st) {
  int rc = ffc_set_lds(big_kernel<N0, DT, FWD>, GeoBig<N0>::LDS_BYTES);
  if (rc) return rc;
  const int64_t nwg = (int64_t)a.npair * a.Hin * (a.Mi / GeoBig<N0>::Mi);
  if (nwg <= 0 || nwg > 2147483647LL) return ffc_fail("outer pass: bad grid");
  hipLaunchKernelGGL((big_kernel<N0, DT, FWD>), dim3((unsigned)nwg), dim3(GeoBig<N0>::WGW * 64), GeoBig<N0>::LDS_BYTES, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : ffc_fail(std::string("big_kernel launch: ") + hipGetErrorString(e));
}

// persistent, double-buffered form (BigBody::run_pipe): one workgroup per CU walks the blocks, the next block's rows arrive by
// LDS-DMA while the current one is transformed and stored
template <int N0, int DT, bool FWD>
__global__ __launch_bounds__((GeoBig<N0>::WGW + 1) * 64, 1) void big_pipe_kernel(BigArgs a) {      // + the loader wave
  BigBody<DevB, N0, DT>::template run_pipe<FWD>(a, blockIdx.x, gridDim.x);
}
template <int N0, int DT, bool FWD>
static int launch_big_pipe(const BigArgs& a, int num_cu, hipStream_t st) {
  using BB = BigBody<DevB, N0, DT>;
  int rc = ffc_set_lds(big_pipe_kernel<N0, DT, FWD>, BB::PIPE_LDS);
  if (rc) return rc;
  const int64_t nblk = (int64_t)a.npair * a.Hin * (a.Mi / GeoBig<N0>::Mi);
  if (nblk <= 0 || nblk > 2147483647LL) return ffc_fail("outer pass: bad grid");
  const unsigned grid = (unsigned)(nblk < num_cu ? nblk : num_cu);
  hipLaunchKernelGGL((big_pipe_kernel<N0, DT, FWD>), dim3(grid), dim3((GeoBig<N0>::WGW + 1) * 64), BB::PIPE_LDS, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : ffc_fail(std::string("big_pipe_kernel launch: ") + hipGetErrorString(e));
}
// The pipelined form is OPT-IN (tuning flag 16, FFC_FLAGS=16) for passes that qualify (16-byte accesses, no input gate on the forward
// side): measured on MI355X it LOSES to run<> -- module level fwd / bwd ms at B16 H768, profiles/r04_ab_bigpipe.txt: fft 262144
// 12.0 / 13.7 against 11.0 / 12.8, fft 1M 49.5 / 55.5 against 47.8 / 53.6 (a first version without the loader wave: 12.6 / 14.5).  Two
// independent one-block workgroups per CU keep more of the HBM pipe busy than one workgroup that double-buffers.  Kept because the
// GPU parity suite ran green on it and the next attempt (two pipelined workgroups of 32 KB blocks) starts from here.
template <int N0, int DT, bool FWD>
static int launch_level(const BigArgs& a, const ffc_plan* p, hipStream_t st) {
  const bool pipe = a.R == 1 && a.fast && !a.lf32 && !a.half && !(FWD && a.gate) && (p->env_flags & 16);
  return pipe ? launch_big_pipe<N0, DT, FWD>(a, p->num_cu, st) : launch_big<N0, DT, FWD>(a, st);
}

// `dtype` of the level entry points: bits 0..3 the 16-bit type; | FFC_LONG_F32 (16): the long side is fp32 (dir = 1: `in` is read as
// float, multiplied by 2^e, e = bits 8..15, and rounded to the 16-bit type; dir = 0: `out` is written as float).  Gates stay 16-bit.
// | FFC_HALF_ROWS (32): the long side is one REAL row per head (Bv == 1) and the short side holds the rows k0 <= K / 2 only (BigArgs::half).
static int decode_dtype(int dtype, int dir, BigArgs* a) {
  a->lf32 = (dtype & 16) ? 1 : 0;
  a->half = (dtype & 32) ? 1 : 0;
  const int e = (dtype >> 8) & 0xff;
  if (e > 30 || (e && !(a->lf32 && dir))) return -1;
  a->lpre = (float)(1u << e);
  return dtype & 15;
}

// One outer level.  dir = 1 forward (long side `in` (Bv,Hin,Llong) -> `out` (2*npair, Hin*N0, Mi)),
// dir = 0 inverse (`in` (2*npair, Hin*N0, Mi) -> long side `out` (Bv,Hin,Llong)).
// `plan` supplies the N0-point DFT operand table in `dtype` (any plan of a size whose outer digit is N0).
extern "C" int ffc_outer_pass(const ffc_plan* plan16, const ffc_plan* plan32, int n0, int dtype, int dir, const void* in, void* out,
                              const void* gate, int64_t Bv, int64_t npair, int64_t Hin, int64_t Mi, int64_t Llong, float scale,
                              void* stream) {
  const ffc_plan* p = n0 == 16 ? plan16 : plan32;
  if (!p || !in || !out) return ffc_fail("null arg");
  if (n0 != p->hp.N1) return ffc_fail("outer pass: plan's outer digit does not match n0");
  if (Mi % (n0 == 16 ? GeoBig<16>::Mi : GeoBig<32>::Mi)) return ffc_fail("outer pass: Mi must be a multiple of the column block");
  if (Llong <= 0 || Llong > n0 * Mi) return ffc_fail("outer pass: bad length");
  BigArgs a{};
  a.in = in; a.out = out; a.gate = gate;
  dtype = decode_dtype(dtype, dir, &a);
  if (dtype != DT_BF16 && dtype != DT_F16) return ffc_fail("outer pass: bad dtype");
  if (a.half && (Bv != 1 || npair != 1)) return ffc_fail("outer pass (half rows): one real row per head only (Bv == 1)");
  const bool bf = dtype == DT_BF16;
  a.fmat = (bf ? p->d_blob_bf : p->d_blob) + (bf ? p->hp_bf.tabs.mat[0] : p->hp.tabs.mat[0]);
  if (!bf && p->hp.dtype != DT_F16) return ffc_fail("outer pass: fp16 tables need an fp16 plan");
  a.Bp_valid = (int)Bv; a.npair = (int)npair; a.Hin = (int)Hin; a.Mi = (int)Mi; a.Llong = (int)Llong; a.scale = scale;
  a.fast = (Llong % 8 == 0) && !(((uintptr_t)in | (uintptr_t)out | (uintptr_t)gate) & 15);
  a.R = 1; a.c = 0;
  hipStream_t st = (hipStream_t)stream;
  if (n0 == 16) {
    if (bf) return dir ? launch_level<16, DT_BF16, true>(a, p, st) : launch_level<16, DT_BF16, false>(a, p, st);
    return dir ? launch_level<16, DT_F16, true>(a, p, st) : launch_level<16, DT_F16, false>(a, p, st);
  }
  if (bf) return dir ? launch_level<32, DT_BF16, true>(a, p, st) : launch_level<32, DT_BF16, false>(a, p, st);
  return dir ? launch_level<32, DT_F16, true>(a, p, st) : launch_level<32, DT_F16, false>(a, p, st);
}

// One level of factor R * 32 as R passes of the 32-point kernel (BigArgs::R): fft 4194304 = 128 x 32768 in ONE HBM level when
// the long side is at most a quarter of it (Llong <= 32 * Mi, the HyenaDNA shape L = N / 4), instead of two levels of 16
// (the reference runs 4M in one level with its 128-point butterfly: csrc/flashfftconv/butterfly/butterfly_padded_cuda_bf16.cu
// :302-487, conv.py:511-551).  `plan_r`: a multi-pass plan with the same R and a 32-point outer digit (fft 131072 for R = 4):
// its per-pass outer-digit tables are the matrices needed here.  dir = 1: every pass reads the long side and writes its
// 32 rows of the R * 32 per head; dir = 0: pass c reads its rows, passes c > 0 add to the long side (call them in order).
// An output gate (dir = 0) multiplies every pass's contribution before it is added: the same product, distributed over the sum.
extern "C" int ffc_outer_pass_r(const ffc_plan* plan_r, int c, int dtype, int dir, const void* in, void* out, const void* gate,
                                int64_t Bv, int64_t npair, int64_t Hin, int64_t Mi, int64_t Llong, float scale, void* stream) {
  const ffc_plan* p = plan_r;
  if (!p || !in || !out) return ffc_fail("null arg");
  if (p->hp.N1 != 32 || p->hp.R < 2) return ffc_fail("outer pass (R passes): needs a multi-pass plan with a 32-point outer digit");
  const int R = p->hp.R;
  if (c < 0 || c >= R) return ffc_fail("outer pass (R passes): bad pass index");
  if (Mi % GeoBig<32>::Mi) return ffc_fail("outer pass: Mi must be a multiple of the column block");
  if (Llong <= 0 || Llong > 32 * Mi) return ffc_fail("outer pass (R passes): the long side must fit the first 32 rows (L <= N / R)");
  BigArgs a{};
  a.in = in; a.out = out; a.gate = gate;
  dtype = decode_dtype(dtype, dir, &a);
  if (dtype != DT_BF16 && dtype != DT_F16) return ffc_fail("outer pass: bad dtype");
  if (a.half && (Bv != 1 || npair != 1)) return ffc_fail("outer pass (half rows): one real row per head only (Bv == 1)");
  const bool bf = dtype == DT_BF16;
  if (!bf && p->hp.dtype != DT_F16) return ffc_fail("outer pass: fp16 tables need an fp16 plan");
  a.fmat = (bf ? p->d_blob_bf : p->d_blob) + (bf ? p->hp_bf.tabs.matk[c][dir ? 0 : 1] : p->hp.tabs.matk[c][dir ? 0 : 1]);
  a.Bp_valid = (int)Bv; a.npair = (int)npair; a.Hin = (int)Hin; a.Mi = (int)Mi; a.Llong = (int)Llong; a.scale = scale;
  a.fast = (Llong % 8 == 0) && !(((uintptr_t)in | (uintptr_t)out | (uintptr_t)gate) & 15);
  a.R = R; a.c = c;
  hipStream_t st = (hipStream_t)stream;
  if (bf) return dir ? launch_big<32, DT_BF16, true>(a, st) : launch_big<32, DT_BF16, false>(a, st);
  return dir ? launch_big<32, DT_F16, true>(a, st) : launch_big<32, DT_F16, false>(a, st);
}

// ---- all R passes of a factor R * 32 in one launch (BigBody::run_all): the forward reads the long side once instead of R times,
// the inverse sums the passes in its fp32 accumulators and writes the long side once (no read-modify-write between launches)
template <int DT, bool FWD>
__global__ __launch_bounds__(GeoBig<32>::WGW * 64, 2) void big_all_kernel(BigArgs a) {
  BigBody<DevB, 32, DT>::template run_all<FWD>(a, blockIdx.x);
}
template <int DT, bool FWD>
static int launch_big_all(const BigArgs& a, hipStream_t st) {
  int rc = ffc_set_lds(big_all_kernel<DT, FWD>, GeoBig<32>::EBYTES);
  if (rc) return rc;
  const int64_t nwg = (int64_t)a.npair * a.Hin * (a.Mi / GeoBig<32>::Mi);
  if (nwg <= 0 || nwg > 2147483647LL) return ffc_fail("outer pass: bad grid");
  hipLaunchKernelGGL((big_all_kernel<DT, FWD>), dim3((unsigned)nwg), dim3(GeoBig<32>::WGW * 64), GeoBig<32>::EBYTES, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : ffc_fail(std::string("big_all_kernel launch: ") + hipGetErrorString(e));
}
// the WIDE form (BigBody::run_wide, round 6): the long side holds up to R * 32 rows -- fft 4194304 = 128 x 32768 and 2097152 = 64 x 32768 in ONE level at
// any length (the reference's 128- / 64-point butterflies take any length too: csrc/flashfftconv/butterfly/butterfly_padded_cuda_bf16.cu:302-487)
template <int DT, bool FWD>
__global__ __launch_bounds__(GeoBig<32>::WGW * 64, 2) void big_wide_kernel(BigArgs a) {
  BigBody<DevB, 32, DT>::template run_wide<FWD>(a, blockIdx.x);
}
template <int DT, bool FWD>
static int launch_big_wide(const BigArgs& a, hipStream_t st) {
  int rc = ffc_set_lds(big_wide_kernel<DT, FWD>, GeoBig<32>::EBYTES);
  if (rc) return rc;
  const int cols = 128 * (GeoBig<32>::WGW / a.R);      // a workgroup's column block: WGW / R groups of 128, R waves (passes / row blocks) on each
  const int64_t nwg = (int64_t)a.npair * a.Hin * (a.Mi / cols);
  if (nwg <= 0 || nwg > 2147483647LL) return ffc_fail("outer pass: bad grid");
  hipLaunchKernelGGL((big_wide_kernel<DT, FWD>), dim3((unsigned)nwg), dim3(GeoBig<32>::WGW * 64), GeoBig<32>::EBYTES, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : ffc_fail(std::string("big_wide_kernel launch: ") + hipGetErrorString(e));
}
// Same level as the R calls ffc_outer_pass_r(plan_r, c = 0 .. R-1, ...), in one launch.
extern "C" int ffc_outer_pass_all(const ffc_plan* plan_r, int dtype, int dir, const void* in, void* out, const void* gate,
                                  int64_t Bv, int64_t npair, int64_t Hin, int64_t Mi, int64_t Llong, float scale, void* stream) {
  const ffc_plan* p = plan_r;
  if (!p || !in || !out) return ffc_fail("null arg");
  if (p->hp.N1 != 32 || p->hp.R < 2 || p->hp.R > 4) return ffc_fail("outer pass (R passes): needs a multi-pass plan with a 32-point outer digit");
  if (Mi % GeoBig<32>::Mi) return ffc_fail("outer pass: Mi must be a multiple of the column block");
  if (Llong <= 0 || Llong > (int64_t)p->hp.R * 32 * Mi) return ffc_fail("outer pass (R passes): the long side is longer than the level (L > N)");
  BigArgs a{};
  a.in = in; a.out = out; a.gate = gate;
  dtype = decode_dtype(dtype, dir, &a);
  if (dtype != DT_BF16 && dtype != DT_F16) return ffc_fail("outer pass: bad dtype");
  if (a.half && (Bv != 1 || npair != 1)) return ffc_fail("outer pass (half rows): one real row per head only (Bv == 1)");
  const bool bf = dtype == DT_BF16;
  if (!bf && p->hp.dtype != DT_F16) return ffc_fail("outer pass: fp16 tables need an fp16 plan");
  for (int c = 0; c < p->hp.R; c++)
    a.fmats[c] = (bf ? p->d_blob_bf : p->d_blob) + (bf ? p->hp_bf.tabs.matk[c][dir ? 0 : 1] : p->hp.tabs.matk[c][dir ? 0 : 1]);
  a.fmat = a.fmats[0];
  a.Bp_valid = (int)Bv; a.npair = (int)npair; a.Hin = (int)Hin; a.Mi = (int)Mi; a.Llong = (int)Llong; a.scale = scale;
  a.fast = (Llong % 8 == 0) && !(((uintptr_t)in | (uintptr_t)out | (uintptr_t)gate) & 15);
  a.R = p->hp.R; a.c = 0;
  hipStream_t st = (hipStream_t)stream;
  if (Llong > 32 * Mi) {      // more than the first 32 long-side rows: the wide form
    if (a.lf32 || a.half) return ffc_fail("outer pass (R passes), long side beyond the first 32 rows: 16-bit rows and all short-side rows only");
    if (GeoBig<32>::WGW % a.R) return ffc_fail("outer pass (R passes), wide form: R must divide the workgroup's waves");
    a.wide = 1;
    if (bf) return dir ? launch_big_wide<DT_BF16, true>(a, st) : launch_big_wide<DT_BF16, false>(a, st);
    return dir ? launch_big_wide<DT_F16, true>(a, st) : launch_big_wide<DT_F16, false>(a, st);
  }
  if (bf) return dir ? launch_big_all<DT_BF16, true>(a, st) : launch_big_all<DT_BF16, false>(a, st);
  return dir ? launch_big_all<DT_F16, true>(a, st) : launch_big_all<DT_F16, false>(a, st);
}
