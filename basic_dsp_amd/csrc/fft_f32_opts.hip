// f32 FFT kernels, the "options" unit (fft_impl.h, BDSP_FFT_PART): fused shift / scale / window / real input / magnitude passes, generic I/O
#define BDSP_FFT_T float
#define BDSP_FFT_PART 2
#include "fft_impl.h"
