// f32 instantiation of the FFT kernels (split from f64 so the two compile in parallel)
#define BDSP_FFT_T float
#define BDSP_FFT_F32_TU 1
#include "fft_impl.h"
