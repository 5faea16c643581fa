// the f64 instantiations of the second-generation block kernel (conv_v2_impl.h)
#define BDSP_CONV_T double
#include "conv_v2_impl.h"
