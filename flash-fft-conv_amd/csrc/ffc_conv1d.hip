// C-ABI entry points of the short depthwise conv1d: dispatch on the input dtype to the three translation
// units that hold the kernels (ffc_conv1d_t{0,1,2}.hip <- ffc_conv1d_impl.h).
#include <stdint.h>

#include "../../include/flashfftconv_hip.h"

extern "C" void ffc_set_error_(const char* m);
extern "C" {
#define FFC_DECL(n)                                                                                                        \
  int ffc_c1d_fwd_##n(const void*, const void*, const void*, void*, int, int, int64_t, int64_t, int64_t, int, int, int, void*); \
  int ffc_c1d_bwd_##n(const void*, const void*, const void*, void*, float*, float*, int, int, int64_t, int64_t, int64_t, int, int, int, void*);
FFC_DECL(0) FFC_DECL(1) FFC_DECL(2)
#undef FFC_DECL

int ffc_conv1d_fwd(const void* u, const void* w, const void* bias, void* y, int in_dtype, int w_dtype, int64_t B, int64_t D,
                   int64_t L, int K, int P, int is_bhl, void* stream) {
  switch (in_dtype) {
    case 0: return ffc_c1d_fwd_0(u, w, bias, y, in_dtype, w_dtype, B, D, L, K, P, is_bhl, stream);
    case 1: return ffc_c1d_fwd_1(u, w, bias, y, in_dtype, w_dtype, B, D, L, K, P, is_bhl, stream);
    case 2: return ffc_c1d_fwd_2(u, w, bias, y, in_dtype, w_dtype, B, D, L, K, P, is_bhl, stream);
  }
  ffc_set_error_("unsupported dtype combination");
  return 1;
}

int ffc_conv1d_bwd(const void* dout, const void* u, const void* w, void* du, float* dw, float* dbias, int in_dtype, int w_dtype,
                   int64_t B, int64_t D, int64_t L, int K, int P, int is_bhl, void* stream) {
  switch (in_dtype) {
    case 0: return ffc_c1d_bwd_0(dout, u, w, du, dw, dbias, in_dtype, w_dtype, B, D, L, K, P, is_bhl, stream);
    case 1: return ffc_c1d_bwd_1(dout, u, w, du, dw, dbias, in_dtype, w_dtype, B, D, L, K, P, is_bhl, stream);
    case 2: return ffc_c1d_bwd_2(dout, u, w, du, dw, dbias, in_dtype, w_dtype, B, D, L, K, P, is_bhl, stream);
  }
  ffc_set_error_("unsupported dtype combination");
  return 1;
}
}
