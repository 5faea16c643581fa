// f64 instantiation of the FFT kernels
#define BDSP_FFT_T double
#include "fft_impl.h"
