// f64 FFT kernels, the "plain" unit (fft_impl.h, BDSP_FFT_PART)
#define BDSP_FFT_T double
#define BDSP_FFT_PART 1
#include "fft_impl.h"
