// conv1d kernels for input dtype 0 (bf16); see ffc_conv1d_impl.h
#define FFC_C1D_TI 0
#include "ffc_conv1d_impl.h"
