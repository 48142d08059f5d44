// libflashfftconv_hip.so: gfx950 device backend for the kernel body + launchers + C-ABI.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>

#define FFC_FN __device__ __forceinline__
#include "../../include/flashfftconv_hip.h"
#include "ffc_body.h"
#include "ffc_modes.h"

namespace ffc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) uint8_t ffc_smem[];

// gfx950 wave backend: every "vector" of the body is one value per lane.
struct DevB {
  using f32 = float;
  using i32 = int;
  using u32 = uint32_t;
  using pred = bool;
  struct U2 { u32 x, y; };
  struct U4 { u32 x, y, z, w; };
  static constexpr bool HAS_TR = true;

  static FFC_FN i32 lane() { return (int)(threadIdx.x & 63); }
  static FFC_FN int wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
  static FFC_FN void barrier() { __syncthreads(); }
  static FFC_FN f32 fconst(float c) { return c; }
  static FFC_FN pred ptrue() { return true; }
  static FFC_FN pred pfalse() { return false; }
  static FFC_FN f32 as_f32(u32 a) { return __builtin_bit_cast(float, a); }
  static FFC_FN u32 as_u32(f32 a) { return __builtin_bit_cast(uint32_t, a); }

  static FFC_FN U2 lds_r64(i32 off) { uint2 v = *(const uint2*)(ffc_smem + off); return U2{v.x, v.y}; }
  static FFC_FN void lds_w64(i32 off, U2 v) { *(uint2*)(ffc_smem + off) = make_uint2(v.x, v.y); }
  static FFC_FN void lds_w32(i32 off, u32 v) { *(uint32_t*)(ffc_smem + off) = v; }
  static FFC_FN void lds_w128(i32 off, U4 v, pred p) {
    if (p) *(uint4*)(ffc_smem + off) = make_uint4(v.x, v.y, v.z, v.w);
  }
  static FFC_FN U4 lds_r128(i32 off) { uint4 v = *(const uint4*)(ffc_smem + off); return U4{v.x, v.y, v.z, v.w}; }
  static FFC_FN u32 lds_r16(i32 off) { return *(const uint16_t*)(ffc_smem + off); }
  static FFC_FN U2 lds_r64_tr(i32 off) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ffc_smem + off));
    uint2 v = __builtin_bit_cast(uint2, t);
    return U2{v.x, v.y};
  }
  static FFC_FN void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  static FFC_FN u32 uconst(uint32_t c) { return c; }
  // hide a value from LICM/CSE so per-phase address math is recomputed instead of kept live
  static FFC_FN i32 opaque(i32 x) { asm volatile("" : "+v"(x)); return x; }
  static FFC_FN u32 sel(pred p, u32 a, u32 b) { return p ? a : b; }
  static FFC_FN u32 g_r16(const void* base, i32 e, pred p) {
    uint16_t v = 0;
    if (p) v = ((const uint16_t*)base)[e];
    return v;
  }
  static FFC_FN void g_w16(void* base, i32 e, u32 v, pred p) {
    if (p) ((uint16_t*)base)[e] = (uint16_t)v;
  }
  static FFC_FN U2 g_r64(const void* base, i32 o8, pred p) {
    uint2 v = make_uint2(0, 0);
    if (p) v = ((const uint2*)base)[o8];
    return U2{v.x, v.y};
  }
  static FFC_FN void g_w64(void* base, i32 o8, U2 v, pred p) {
    if (p) ((uint2*)base)[o8] = make_uint2(v.x, v.y);
  }
  static FFC_FN U4 g_r128(const void* base, i32 o16) {
    uint4 v = ((const uint4*)base)[o16];
    return U4{v.x, v.y, v.z, v.w};
  }
  static FFC_FN U4 g_r128p(const void* base, i32 o16, pred p) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p) v = ((const uint4*)base)[o16];
    return U4{v.x, v.y, v.z, v.w};
  }
  static FFC_FN void g_w128(void* base, i32 o16, U4 v, pred p) {
    if (p) ((uint4*)base)[o16] = make_uint4(v.x, v.y, v.z, v.w);
  }
  template <int DT>
  static FFC_FN void mfma(f32 (&acc)[16], const u32 (&a)[4], const u32 (&b)[4]) {
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = acc[i];
    u32x4v av = {a[0], a[1], a[2], a[3]}, bv = {b[0], b[1], b[2], b[3]};
    if (DT == DT_BF16)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = c[i];
  }
  template <int DT>
  static FFC_FN u32 pack(f32 lo, f32 hi) {
    f32x2 v = {lo, hi};
    if (DT == DT_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  template <int DT>
  static FFC_FN f32 unpack_lo(u32 a) {
    if (DT == DT_BF16) return __builtin_bit_cast(float, a << 16);
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(a & 0xffffu));
  }
  template <int DT>
  static FFC_FN f32 unpack_hi(u32 a) {
    if (DT == DT_BF16) return __builtin_bit_cast(float, a & 0xffff0000u);
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(a >> 16));
  }
};

// blockIdx -> (head, chunk).  Blocks land on XCD (id % 8): keep all chunks of one head on one XCD
// so k_f[h] is served by that XCD's L2 (speed only, never correctness).
__device__ __forceinline__ bool map_block(int H, int nchunk, int* h, int* chunk) {
  int id = blockIdx.x;
  int xcd = id & 7, s = id >> 3;
  *h = xcd + 8 * (s / nchunk);
  *chunk = s % nchunk;
  return *h < H;
}

template <class GEO, int DT>
__global__ __launch_bounds__(GEO::WGW * 64, 2) void conv_kernel(ConvArgs a) {
  int h, chunk;
  if (!map_block(a.H, a.nchunk, &h, &chunk)) return;
  Body<DevB, GEO, DT>::conv(a, h, chunk);
}

// k_f natural (H,N) complex64 -> internal order, scaled, dtype.
template <int DT>
__global__ void kf_pack_kernel(const float2* __restrict__ src, const int32_t* __restrict__ freq, uint32_t* __restrict__ dst,
                               int N, int per_h, float scale) {
  int h = blockIdx.y;
  int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= per_h) return;
  float2 v = src[(int64_t)h * N + freq[pos]];
  dst[(int64_t)h * per_h + pos] = DevB::pack<DT>(v.x * scale, v.y * scale);
}

__global__ void selftest_kernel(const uint32_t* in, uint32_t* out) {
  int l = threadIdx.x;
  uint32_t a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = in[l * 8 + i]; b[i] = in[l * 8 + 4 + i]; }
  float acc[16];
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  DevB::mfma<DT_BF16>(acc, a, b);
  for (int i = 0; i < 16; i++) out[l * 40 + i] = DevB::as_u32(acc[i]);
  float acc2[16];
  for (int i = 0; i < 16; i++) acc2[i] = 0.f;
  DevB::mfma<DT_F16>(acc2, a, b);
  for (int i = 0; i < 16; i++) out[l * 40 + 16 + i] = DevB::as_u32(acc2[i]);
  DevB::lds_w64(l * 8, DevB::U2{a[0], a[1]});
  __syncthreads();
  // permuted addresses so the test is not symmetric
  DevB::U2 t = DevB::lds_r64_tr(((l * 5 + 3) & 63) * 8);
  out[l * 40 + 32] = t.x;
  out[l * 40 + 33] = t.y;
  out[l * 40 + 34] = DevB::pack<DT_BF16>(acc[0], acc[1]);
  out[l * 40 + 35] = DevB::pack<DT_F16>(acc[0], acc[1]);
  out[l * 40 + 36] = DevB::as_u32(DevB::unpack_lo<DT_F16>(a[2]));
  out[l * 40 + 37] = DevB::as_u32(DevB::unpack_hi<DT_F16>(a[2]));
  out[l * 40 + 38] = DevB::as_u32(DevB::unpack_lo<DT_BF16>(a[2]));
  out[l * 40 + 39] = DevB::as_u32(DevB::unpack_hi<DT_BF16>(a[2]));
}

}  // namespace ffc

using namespace ffc;

struct ffc_plan {
  HostPlan hp;
  uint8_t* d_blob = nullptr;
  int32_t* d_freq = nullptr;
  int num_cu = 256;
};

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x)                                                                 \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class GEO, int DT>
struct ConvLaunch {
  static int run(const ConvArgs& a, hipStream_t st) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
      attr_err = hipFuncSetAttribute((const void*)conv_kernel<GEO, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, GEO::LDS_BYTES);
    });
    if (attr_err != hipSuccess) return fail(std::string("hipFuncSetAttribute: ") + hipGetErrorString(attr_err));
    int hpad = (a.H + 7) & ~7;
    dim3 grid(hpad * a.nchunk), block(GEO::WGW * 64);
    hipLaunchKernelGGL((conv_kernel<GEO, DT>), grid, block, GEO::LDS_BYTES, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("conv_kernel launch: ") + hipGetErrorString(e));
    return 0;
  }
};

template <template <class, int> class FN, class... A>
static int dispatch(int N, int dtype, A&&... args) {
#define FFC_CASE(NN, a, b, c)                                                   \
  case NN:                                                                      \
    return dtype == DT_BF16 ? FN<Geo<a, b, c>, DT_BF16>::run(args...) : FN<Geo<a, b, c>, DT_F16>::run(args...);
  switch (N) {
    FFC_CASE(256, 1, 16, 16)
    FFC_CASE(512, 1, 16, 32)
    FFC_CASE(1024, 1, 32, 32)
    FFC_CASE(4096, 16, 16, 16)
    FFC_CASE(8192, 32, 16, 16)
    FFC_CASE(16384, 16, 32, 32)
    FFC_CASE(32768, 32, 32, 32)
  }
#undef FFC_CASE
  return fail("unsupported fft size");
}

// pairs per chunk / number of chunks so the grid fills the chip (>= ~2 waves of workgroups) while a
// workgroup still loops over several pairs of one head (k_f[h] reuse through L2).
static void choose_chunks(const ffc_plan* p, int H, int npair, int* nchunk, int* ppc) {
  const bool outer = p->hp.N1 > 1;
  int upw = 8 / p->hp.NW;                       // units a workgroup processes per iteration
  int pairs_per_iter = outer ? upw : upw * p->hp.G;
  int wg_per_cu = outer ? 1 : 2;
  int target = p->num_cu * wg_per_cu * 2;
  int iters_total = (npair + pairs_per_iter - 1) / pairs_per_iter;
  int nc = (target + H - 1) / H;
  if (nc > iters_total) nc = iters_total;
  if (nc < 1) nc = 1;
  int ipc = (iters_total + nc - 1) / nc;
  *ppc = ipc * pairs_per_iter;
  *nchunk = (npair + *ppc - 1) / *ppc;
}

extern "C" {

int ffc_version(void) { return 100; }
const char* ffc_last_error(void) { return g_err.c_str(); }

int ffc_plan_create(int64_t fft_size, int dtype, ffc_plan** out) {
  if (!out) return fail("null out");
  ffc_plan* p = new ffc_plan();
  if (!build_plan((int)fft_size, dtype, &p->hp)) {
    delete p;
    return fail("unsupported fft_size/dtype (supported: 256,512,1024,4096,8192,16384,32768; bf16/fp16)");
  }
  hipError_t e = hipMalloc((void**)&p->d_blob, p->hp.blob.size());
  if (e == hipSuccess) e = hipMemcpy(p->d_blob, p->hp.blob.data(), p->hp.blob.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_freq, p->hp.kf_freq.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(p->d_freq, p->hp.kf_freq.data(), p->hp.kf_freq.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    std::string m = std::string("plan upload: ") + hipGetErrorString(e);
    ffc_plan_destroy(p);
    return fail(m);
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) p->num_cu = prop.multiProcessorCount;
  *out = p;
  return 0;
}

void ffc_plan_destroy(ffc_plan* p) {
  if (!p) return;
  if (p->d_blob) (void)hipFree(p->d_blob);
  if (p->d_freq) (void)hipFree(p->d_freq);
  delete p;
}

int64_t ffc_plan_kf_elems(const ffc_plan* p) { return p ? (int64_t)p->hp.NT * 1024 : 0; }
double ffc_plan_kf_scale(const ffc_plan* p) { return p ? p->hp.s_k : 0; }
int ffc_plan_kf_index(const ffc_plan* p, int32_t* out) {
  if (!p || !out) return fail("null arg");
  memcpy(out, p->hp.kf_freq.data(), p->hp.kf_freq.size() * 4);
  return 0;
}

int ffc_kf_pack(const ffc_plan* p, const void* src, int64_t H, void* dst, void* stream) {
  if (!p || !src || !dst) return fail("null arg");
  int per_h = p->hp.NT * 1024;
  dim3 grid((per_h + 255) / 256, (unsigned)H), block(256);
  if (p->hp.dtype == DT_BF16)
    hipLaunchKernelGGL(kf_pack_kernel<DT_BF16>, grid, block, 0, (hipStream_t)stream, (const float2*)src, p->d_freq,
                       (uint32_t*)dst, p->hp.N, per_h, (float)p->hp.s_k);
  else
    hipLaunchKernelGGL(kf_pack_kernel<DT_F16>, grid, block, 0, (hipStream_t)stream, (const float2*)src, p->d_freq,
                       (uint32_t*)dst, p->hp.N, per_h, (float)p->hp.s_k);
  HIPCHK(hipGetLastError());
  return 0;
}

int ffc_conv_fwd(const ffc_plan* p, const void* u, const void* kf, const void* pregate, const void* postgate, void* y,
                 int64_t B, int64_t H, int64_t L, int conj_kf, void* stream) {
  if (!p || !u || !kf || !y) return fail("null arg");
  if (B <= 0 || H <= 0) return fail("empty batch/heads");
  if (L <= 0 || L > p->hp.N) return fail("L must be in (0, fft_size]");
  if ((uintptr_t)kf & 15) return fail("k_f must be 16-byte aligned");
  if (B * H * L >= ((int64_t)1 << 31)) return fail("tensor too large (>= 2^31 elements)");
  ConvArgs a{};
  a.u = u; a.pregate = pregate; a.postgate = postgate; a.y = y; a.kf = kf;
  a.tab = p->d_blob; a.t = p->hp.tabs;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.npair = (int)((B + 1) / 2);
  a.conj_kf = conj_kf;
  a.fast = (L % 8 == 0) && !(((uintptr_t)u | (uintptr_t)y | (uintptr_t)pregate | (uintptr_t)postgate) & 15);
  choose_chunks(p, a.H, a.npair, &a.nchunk, &a.ppc);
  return dispatch<ConvLaunch>(p->hp.N, p->hp.dtype, a, (hipStream_t)stream);
}

int ffc_kernel_fft(const ffc_plan*, const float*, int64_t, int64_t, void*, void*) { return fail("ffc_kernel_fft: not implemented yet"); }
int64_t ffc_dkf_workspace_bytes(const ffc_plan*, int64_t, int64_t) { return 0; }
int ffc_conv_bwd_dkf(const ffc_plan*, const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, void*) {
  return fail("ffc_conv_bwd_dkf: not implemented yet");
}
int ffc_kernel_ifft_grad(const ffc_plan*, const void*, int64_t, int64_t, int64_t, float*, void*) {
  return fail("ffc_kernel_ifft_grad: not implemented yet");
}
int ffc_conv1d_fwd(const void*, const void*, const void*, void*, int, int, int64_t, int64_t, int64_t, int, int, int, void*) {
  return fail("ffc_conv1d_fwd: not implemented yet");
}
int ffc_conv1d_bwd(const void*, const void*, const void*, void*, float*, float*, int, int, int64_t, int64_t, int64_t, int,
                   int, int, void*) {
  return fail("ffc_conv1d_bwd: not implemented yet");
}

int ffc_selftest_primitives(const uint32_t* in_host, uint32_t* out_host) {
  uint32_t *din = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc((void**)&din, 64 * 8 * 4));
  HIPCHK(hipMalloc((void**)&dout, 64 * 40 * 4));
  HIPCHK(hipMemcpy(din, in_host, 64 * 8 * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 1024, 0, din, dout);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_host, dout, 64 * 40 * 4, hipMemcpyDeviceToHost));
  (void)hipFree(din);
  (void)hipFree(dout);
  return 0;
}

}  // extern "C"
