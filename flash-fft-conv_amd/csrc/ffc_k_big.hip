// HBM-level outer DFT passes for FFT sizes >= 65536 (ffc_big.h) + C-ABI.
#include "ffc_dev.h"
using namespace ffc;

template <int N0, int DT, bool FWD>
__global__ __launch_bounds__(GeoBig<N0>::WGW * 64, 2) void big_kernel(BigArgs a) {
  BigBody<DevB, N0, DT>::template run<FWD>(a, blockIdx.x);
}

template <int N0, int DT, bool FWD>
static int launch_big(const BigArgs& a, hipStream_t st) {
  static int rc = ffc_set_lds(big_kernel<N0, DT, FWD>, GeoBig<N0>::LDS_BYTES);
  if (rc) return rc;
  const int64_t nwg = (int64_t)a.npair * a.Hin * (a.Mi / GeoBig<N0>::Mi);
  if (nwg <= 0 || nwg > 2147483647LL) return ffc_fail("outer pass: bad grid");
  hipLaunchKernelGGL((big_kernel<N0, DT, FWD>), dim3((unsigned)nwg), dim3(GeoBig<N0>::WGW * 64), GeoBig<N0>::LDS_BYTES, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : ffc_fail(std::string("big_kernel launch: ") + hipGetErrorString(e));
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
  const bool bf = dtype == DT_BF16;
  a.fmat = (bf ? p->d_blob_bf : p->d_blob) + (bf ? p->hp_bf.tabs.mat[0] : p->hp.tabs.mat[0]);
  if (!bf && p->hp.dtype != DT_F16) return ffc_fail("outer pass: fp16 tables need an fp16 plan");
  a.Bp_valid = (int)Bv; a.npair = (int)npair; a.Hin = (int)Hin; a.Mi = (int)Mi; a.Llong = (int)Llong; a.scale = scale;
  a.fast = (Llong % 8 == 0) && !(((uintptr_t)in | (uintptr_t)out | (uintptr_t)gate) & 15);
  hipStream_t st = (hipStream_t)stream;
  if (n0 == 16) {
    if (bf) return dir ? launch_big<16, DT_BF16, true>(a, st) : launch_big<16, DT_BF16, false>(a, st);
    return dir ? launch_big<16, DT_F16, true>(a, st) : launch_big<16, DT_F16, false>(a, st);
  }
  if (bf) return dir ? launch_big<32, DT_BF16, true>(a, st) : launch_big<32, DT_BF16, false>(a, st);
  return dir ? launch_big<32, DT_F16, true>(a, st) : launch_big<32, DT_F16, false>(a, st);
}
