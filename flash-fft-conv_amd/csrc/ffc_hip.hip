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
#include <map>
#include <mutex>
int ffc_set_lds_once(const void* kernel, int bytes) {
  // fast path in front of the mutex (every small-size launch paid a lock and a map lookup): the last
  // (kernel, device) pair this thread has seen set
  static thread_local const void* last_k = nullptr;
  static thread_local int last_dev = -1, last_bytes = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (kernel == last_k && dev == last_dev && bytes <= last_bytes) return 0;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;      // (kernel, device) -> largest size set
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find({kernel, dev});
  if (it == done.end() || it->second < bytes) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return ffc_fail(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    done[{kernel, dev}] = bytes;
    it = done.find({kernel, dev});
  }
  last_k = kernel; last_dev = dev; last_bytes = it->second;
  return 0;
}
#define fail ffc_fail

// Test support: leave every CU's LDS and register files full of NaN patterns, so that a kernel which reads LDS or
// registers it never initialised (benign when the previous workgroup on the CU was the same kernel) fails loudly.
extern __shared__ uint32_t ffc_poison_smem[];
__global__ __launch_bounds__(256) void poison_kernel(uint32_t pat, int words, uint32_t* sink) {
  for (int i = threadIdx.x; i < words; i += blockDim.x) ffc_poison_smem[i] = pat;
  asm volatile(".irp r,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n v_mov_b32 v\\r, %0\n .endr" ::"v"(pat) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
  asm volatile(".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n v_accvgpr_write_b32 a\\r, %0\n .endr" ::"v"(pat) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
  __syncthreads();
  if (sink && ffc_poison_smem[(threadIdx.x * 7) % words] != pat) *sink = 1;
  __builtin_amdgcn_s_sleep(64);
}

// Measured peaks of the box the process runs on (bench.py `peak_measured`; SURVEY.md section 6 asks for them next to the
// nominal 8 TB/s / 2.5 PFLOP/s): a 16-byte-per-lane stream copy and a register-resident v_mfma_f32_32x32x16_bf16 loop.
// 4 x 16 bytes per lane and iteration, streaming (non-temporal) loads and stores: the lines are touched once
__global__ __launch_bounds__(256) void peak_copy_kernel(const u32x4v* __restrict__ src, u32x4v* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    u32x4v a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    u32x4v c = __builtin_nontemporal_load(src + i + 2 * stride), e = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(e, dst + i + 3 * stride);
  }
  for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ __launch_bounds__(256) void peak_mfma_kernel(float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  DevB::A16 acc[4];
  for (int j = 0; j < 4; j++) acc[j] = DevB::a16_zero();
  DevB::W4 a = DevB::w4(0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), b = DevB::w4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) DevB::mfma<DT_BF16>(acc[j], a, b);
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
  if (s == 12345.678f) sink[0] = s;
}

extern "C" {

void ffc_set_error_(const char* m) { g_err = m; }
int ffc_debug_peaks(double* copy_GBs, double* mfma_TFLOPs) {
  if (!copy_GBs || !mfma_TFLOPs) return fail("null arg");
  int dev = 0, ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
  // 2 GiB each way (well past the 256 MB Infinity Cache), on a stream of its own; everything is released on every path
  const size_t bytes = (size_t)2 << 30;
  u32x4v *s = nullptr, *d = nullptr; float* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
  bool ok = hipMalloc((void**)&s, bytes) == hipSuccess && hipMalloc((void**)&d, bytes) == hipSuccess &&
            hipMalloc((void**)&sink, 64) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
  double bc = 0, bm = 0;
  if (ok) {
    (void)hipMemsetAsync(s, 1, bytes, st); (void)hipMemsetAsync(d, 2, bytes, st);
    // grid sweep (workgroups per CU), best of 3 timed launches each after one warm-up
    const int per_cu[] = {2, 4, 8, 16, 32};
    for (int g : per_cu) {
      for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(peak_copy_kernel, dim3(ncu * g), dim3(256), 0, st, s, d, bytes / 16);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms > 0 && 2.0 * bytes / (ms * 1e-3) / 1e9 > bc) bc = 2.0 * bytes / (ms * 1e-3) / 1e9;
      }
    }
    const int iters = 20000;
    for (int rep = 0; rep < 4; rep++) {
      (void)hipEventRecord(e0, st);
      hipLaunchKernelGGL(peak_mfma_kernel, dim3(ncu * 4), dim3(256), 0, st, sink, iters);
      (void)hipEventRecord(e1, st);
      (void)hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      double fl = (double)ncu * 4 * 4 * iters * 4 * (2.0 * 32 * 32 * 16);
      if (rep && ms > 0 && fl / (ms * 1e-3) / 1e12 > bm) bm = fl / (ms * 1e-3) / 1e12;
    }
    ok = hipGetLastError() == hipSuccess;
  }
  if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  if (s) (void)hipFree(s);
  if (d) (void)hipFree(d);
  if (sink) (void)hipFree(sink);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (!ok) return fail("peak probe: allocation or launch failed");
  *copy_GBs = bc; *mfma_TFLOPs = bm;
  return 0;
}
// (ffc_conv_fwd_k / ffc_conv_bwd_k, the module's one call per direction, live next to the launchers they drive: ffc_k_conv.hip,
// ffc_k_bwd.hip)
int ffc_version(void) { return 102; }
const char* ffc_last_error(void) { return g_err.c_str(); }

// (Re-)read the tuning knobs from the environment into the plan.  Called once by ffc_plan_create; A/B tuning scripts call
// it again after changing a variable.  Launches never call getenv.
void ffc_plan_reload_env(ffc_plan* p) {
  if (!p) return;
  p->env_flags = 0; p->env_stream = -1; p->env_persist = -1; p->env_wg_mult = 0;
  if (const char* e = getenv("FFC_FLAGS")) p->env_flags = atoi(e);
  if (const char* e = getenv("FFC_STREAM")) p->env_stream = atoi(e) ? 1 : 0;
  if (const char* e = getenv("FFC_PERSIST")) p->env_persist = atoi(e) > 0 ? atoi(e) : 0;
  if (const char* e = getenv("FFC_WG_MULT")) p->env_wg_mult = atoi(e);
}

int ffc_plan_create(int64_t fft_size, int dtype, ffc_plan** out) {
  if (!out) return fail("null out");
  ffc_plan* p = new ffc_plan();
  if (!build_plan((int)fft_size, dtype, &p->hp)) {
    delete p;
    return fail("unsupported fft_size/dtype (supported: 256,512,1024,2048,4096,8192,16384,32768,65536,131072; bf16/fp16)");
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
  ffc_plan_reload_env(p);
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

int64_t ffc_plan_kf_elems(const ffc_plan* p) { return p ? (int64_t)p->hp.R * p->hp.NT * 1024 : 0; }
double ffc_plan_kf_scale(const ffc_plan* p) { return p ? p->hp.s_k : 0; }
int ffc_plan_kf_index(const ffc_plan* p, int32_t* out) {
  if (!p || !out) return fail("null arg");
  memcpy(out, p->hp.kf_freq.data(), p->hp.kf_freq.size() * 4);
  return 0;
}

int ffc_kf_pack(const ffc_plan* p, const void* src, int64_t H, void* dst, void* stream) {
  if (!p || !src || !dst) return fail("null arg");
  int per_h = p->hp.R * p->hp.NT * 1024;
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

int ffc_debug_poison(void* stream) {
  const int bytes = 160 * 1024;
  HIPCHK(hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  hipLaunchKernelGGL(poison_kernel, dim3(2048), dim3(256), bytes, (hipStream_t)stream, 0x7FC07FC0u, bytes / 4, (uint32_t*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
