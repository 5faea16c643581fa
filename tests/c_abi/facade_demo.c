/* Plain-C client of libbasic_dsp_hip.so: exercises the B1 trait functions and the B2 facade exactly the
 * way a C caller of the reference's interop crate would (interop/src/facade32.rs), with no Python and no
 * C++ in between.  Built and run by tests/test_gpu_parity.py::test_c_client_of_the_abi.
 *   gcc -std=c11 -O1 -I include tests/c_abi/facade_demo.c -L basic_dsp_amd/lib -lbasic_dsp_hip -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "basic_dsp_hip.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL: %s (%s)\n", msg, bdsp_hip_last_error()); return 1; } } while (0)

int main(void)
{
    CHECK(bdsp_hip_has_gpu_support_f32(), "has_gpu_support");
    /* B1: in-place FFT of a host slice, the GpuSupport::fft call (gpu_support/mod.rs:35) */
    enum { N = 16384 };
    float *sig = malloc(sizeof(float) * 2 * N);
    for (int i = 0; i < N; ++i) { sig[2 * i] = cosf(2.0f * (float)M_PI * 5.0f * (float)i / N); sig[2 * i + 1] = 0.0f; }
    CHECK(bdsp_hip_is_supported_fft_len_f32(1, 2 * N), "is_supported_fft_len");
    CHECK(bdsp_hip_fft_f32(1, sig, 2 * N, 0) == 0, "bdsp_hip_fft_f32");
    CHECK(fabsf(sig[2 * 5] - N / 2.0f) < 1e-2f * N && fabsf(sig[2 * 7]) < 1e-2f * N, "tone lands in bin 5");

    /* B2: the facade -- ownership moves in and comes back in the result struct */
    VecBuf32 *v = new32(1, 0, 0.0f, 2 * N, 1.0f);
    CHECK(v != NULL, "new32");
    float *host = malloc(sizeof(float) * 2 * N);
    for (int i = 0; i < 2 * N; ++i) host[i] = (float)(i % 17) - 8.0f;
    VectorInteropResult32 r = overwrite_data32(v, host, 2 * N);
    CHECK(r.result_code == 0, "overwrite_data32");
    r = real_scale32(r.vector, 2.0f);              CHECK(r.result_code == 0, "real_scale32");
    r = plain_fft32(r.vector);                     CHECK(r.result_code == 0, "plain_fft32");
    CHECK(get_domain32(r.vector) == 1 && get_delta32(r.vector) == (float)N, "domain/delta after fft");
    r = plain_ifft32(r.vector);                    CHECK(r.result_code == 0, "plain_ifft32");
    r = real_scale32(r.vector, 0.5f / (float)N);   CHECK(r.result_code == 0, "real_scale32 (normalise)");
    const float *back = data32(r.vector);
    double err = 0, ref = 0;
    for (int i = 0; i < 2 * N; ++i) { err += (back[i] - host[i]) * (double)(back[i] - host[i]); ref += host[i] * (double)host[i]; }
    CHECK(sqrt(err / ref) < 2e-6, "fft/ifft round trip");
    /* error codes are the reference's: a frequency-domain operation on a time vector poisons it (-1) */
    r = plain_ifft32(r.vector);
    CHECK(r.result_code == -1 && get_len32(r.vector) == 0, "poisoned vector reports -1");
    delete_vector32(r.vector);

    /* convolve_signal32 with a borrowed operand */
    VecBuf32 *x = new32(0, 0, 1.0f, 1000, 1.0f), *h = new32(0, 0, 0.25f, 4, 1.0f);
    r = convolve_signal32(x, h);
    CHECK(r.result_code == 0, "convolve_signal32");
    CHECK(fabsf(get_value32(r.vector, 500) - 1.0f) < 1e-5f, "moving average of ones is one");
    delete_vector32(r.vector);
    delete_vector32(h);

    /* statistics / math family / running sums: struct returns by value, and the facade names that collide with
     * <math.h> (included above) through their bdsp_ declarations */
    VecBuf32 *w = new32(0, 0, 2.0f, 1000, 1.0f);
    Statistics32 st = real_statistics32(w);
    CHECK(st.count == 1000 && fabsf(st.sum - 2000.0f) < 1e-3f && st.min == 2.0f && st.max_index == 0, "real_statistics32");
    r = bdsp_powf32(w, 3.0f);                     CHECK(r.result_code == 0, "powf32");
    CHECK(fabsf(get_value32(r.vector, 7) - 8.0f) < 1e-5f, "2^3");
    r = cum_sum32(r.vector);                      CHECK(r.result_code == 0, "cum_sum32");
    CHECK(fabsf(get_value32(r.vector, 999) - 8000.0f) < 1e-2f, "running sum of 1000 eights");
    r = sqrt32(r.vector);                         CHECK(r.result_code == 0, "sqrt32");
    ScalarInteropResult32 dp = real_dot_product32(r.vector, r.vector);
    CHECK(dp.result_code == 0 && fabsf(dp.result - 8.0f * 500500.0f) < 1.0f, "dot product of sqrt(8k) with itself");
    r = plain_fft32(r.vector);                    CHECK(r.result_code == 0, "1000-point (2^3 5^3) real-input fft");
    CHECK(get_len32(r.vector) == 2000, "real -> complex");
    delete_vector32(r.vector);
    free(sig); free(host);
    printf("c abi demo ok (%s)\n", bdsp_hip_version());
    return 0;
}
