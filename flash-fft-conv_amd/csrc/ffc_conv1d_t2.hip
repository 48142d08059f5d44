// conv1d kernels for input dtype 2 (fp32); see ffc_conv1d_impl.h
#define FFC_C1D_TI 2
#include "ffc_conv1d_impl.h"
