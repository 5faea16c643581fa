// f64 FFT kernels, the "options" unit (fft_impl.h, BDSP_FFT_PART)
#define BDSP_FFT_T double
#define BDSP_FFT_PART 2
#include "fft_impl.h"
