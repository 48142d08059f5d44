// Fused backward kernels on the spectra saved by the forward pass (bwd_kernel<.., ZM = 1>, bwd_rp_kernel<.., ZM = 1>):
// see ffc_bwd_launch.h.  Entry points stay in ffc_k_bwd.hip (ffc_conv_bwd_z / ffc_conv_bwd_zy).
#include "ffc_bwd_launch.h"

int ffc_bwdz_launch(int N, int dtype, const ffc::DkfArgs& d, hipStream_t st) {
  return ffc_dispatch<BwdLaunchZ<1>::T>(N, dtype, d, st);
}
