// conv1d kernels for input dtype 1 (fp16); see ffc_conv1d_impl.h
#define FFC_C1D_TI 1
#include "ffc_conv1d_impl.h"
