// f32 FFT kernels, the "plain" unit (fft_impl.h, BDSP_FFT_PART): plan selection + every kernel a plain transform launches
#define BDSP_FFT_T float
#define BDSP_FFT_F32_TU 1
#define BDSP_FFT_PART 1
#include "fft_impl.h"
