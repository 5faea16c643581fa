// the f32 instantiations of the second-generation block kernel (conv_v2_impl.h) + its precision-independent helpers
#define BDSP_CONV_T float
#define BDSP_CONV_F32_TU 1
#include "conv_v2_impl.h"
