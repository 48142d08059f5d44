// libflashfftconv_hip.so core: plan management, k_f packing, primitive self test, error state.
#include "ffc_dev.h"

namespace ffc {

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
  DevB::W4 a, b;
  for (int i = 0; i < 4; i++) { a[i] = in[l * 8 + i]; b[i] = in[l * 8 + 4 + i]; }
  DevB::A16 acc = DevB::a16_zero();
  DevB::mfma<DT_BF16>(acc, a, b);
  for (int i = 0; i < 16; i++) out[l * 40 + i] = DevB::as_u32(acc[i]);
  DevB::A16 acc2 = DevB::a16_zero();
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

static thread_local std::string g_err;
#define fail ffc_fail

extern "C" {

void ffc_set_error_(const char* m) { g_err = m; }
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
  if (dtype == DT_BF16) {
    p->hp_bf = p->hp;
    p->d_blob_bf = p->d_blob;
  } else {
    build_plan((int)fft_size, DT_BF16, &p->hp_bf);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_blob_bf, p->hp_bf.blob.size());
    if (e == hipSuccess) e = hipMemcpy(p->d_blob_bf, p->hp_bf.blob.data(), p->hp_bf.blob.size(), hipMemcpyHostToDevice);
  }
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
  if (p->d_blob_bf && p->d_blob_bf != p->d_blob) (void)hipFree(p->d_blob_bf);
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
