/*
 * basic_dsp_hip.h -- C ABI of libbasic_dsp_hip.so, the MI355X (gfx950) backend for basic_dsp's
 * time/frequency-domain vector operations.
 *
 * Three layers, all plain C (pointers + sizes, no C++/torch types):
 *
 *   B1  bdsp_hip_*_f32/_f64       the five functions of the reference's GPU plug-in trait
 *                                 `GpuSupport<T>` (vector/src/gpu_support/mod.rs:18-46); HOST
 *                                 pointers, synchronous, thread-safe.  This is what a Rust shim
 *                                 `impl GpuSupport<T> for T` binds.
 *   B2  new32, plain_fft32, ...   the subset of the reference's C facade
 *                                 (interop/src/facade32.rs, facade64.rs) that covers the hot path,
 *                                 with IDENTICAL names, argument order and result codes, but the
 *                                 vector lives in HBM.  Handles are opaque.
 *   B3  bdsp_hip_dev_*            the same kernels on caller-owned DEVICE pointers and a caller
 *                                 stream (used by bench.py and the multi-GPU batch driver, which
 *                                 hold memory in torch tensors and communicate through RCCL).
 *
 * All citations are relative to the reference tree (liebharc/basic_dsp v0.10.0).
 * Conventions: complex data is interleaved [re0, im0, re1, im1, ...];
 * every `len` counts SCALARS; `points` = len/2 for complex vectors.
 */
#ifndef BASIC_DSP_HIP_H
#define BASIC_DSP_HIP_H

#include <stddef.h>
#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Result codes.  0 = ok; 1..14 = the reference's ErrorReason numbering
 * (interop/src/lib.rs:125-142, translate_error); -1 = vector is poisoned
 * (interop/src/lib.rs:145-151); <= -100 = backend failure (no reference counterpart: the OpenCL
 * backend panics instead, vector/src/gpu_support/ocl/mod.rs:167,188,208).
 * ---------------------------------------------------------------------------------------- */
#define BDSP_OK 0
#define BDSP_ERR_SAME_SIZE 1             /* InputMustHaveTheSameSize */
#define BDSP_ERR_META_DATA 2             /* InputMetaDataMustAgree */
#define BDSP_ERR_MUST_BE_COMPLEX 3       /* InputMustBeComplex */
#define BDSP_ERR_MUST_BE_REAL 4          /* InputMustBeReal */
#define BDSP_ERR_MUST_BE_TIME 5          /* InputMustBeInTimeDomain */
#define BDSP_ERR_MUST_BE_FREQ 6          /* InputMustBeInFrequencyDomain */
#define BDSP_ERR_ARG_LENGTH 7            /* InvalidArgumentLength */
#define BDSP_ERR_CONJ_SYMMETRIC 8        /* InputMustBeConjSymmetric */
#define BDSP_ERR_ODD_LENGTH 9            /* InputMustHaveAnOddLength */
#define BDSP_ERR_FN_SYMMETRIC 10         /* ArgumentFunctionMustBeSymmetric */
#define BDSP_ERR_COMBINED_ARGS 11        /* InvalidNumberOfArgumentsForCombinedOp */
#define BDSP_ERR_NOT_EMPTY 12            /* InputMustNotBeEmpty */
#define BDSP_ERR_EVEN_LENGTH 13          /* InputMustHaveAnEvenLength */
#define BDSP_ERR_CANNOT_RESIZE 14        /* TypeCanNotResize */
#define BDSP_ERR_POISONED (-1)
#define BDSP_ERR_NO_DEVICE (-100)        /* no usable gfx950 device / HIP runtime failure */
#define BDSP_ERR_HIP (-101)              /* a HIP call failed; see bdsp_hip_last_error() */
#define BDSP_ERR_UNSUPPORTED (-102)      /* argument combination this backend does not implement */

/* Human-readable text of the last backend failure on the calling thread ("" if none). */
const char *bdsp_hip_last_error(void);
/* Library version string, e.g. "basic_dsp_hip 0.1.0 (gfx950)". */
const char *bdsp_hip_version(void);

/* ==========================================================================================
 * B1 -- GpuSupport<T> (vector/src/gpu_support/mod.rs:18-46).  HOST pointers.
 * ======================================================================================== */

/* fn has_gpu_support() -> bool            (gpu_support/mod.rs:21)
 * 1 if a gfx950 device is usable (for _f64: always the same answer, MI355X has native fp64). */
int bdsp_hip_has_gpu_support_f32(void);
int bdsp_hip_has_gpu_support_f64(void);

/* fn is_supported_fft_len(is_complex, len) -> bool      (gpu_support/mod.rs:32)
 * `len` in scalars.  Like the OpenCL backend (ocl/mod.rs:277-299) real input is refused; unlike
 * it, EVERY complex length is supported (powers of two and 2,3,5,7-smooth lengths natively, all other
 * lengths through Bluestein on the same kernels), matching what rustfft accepts on the CPU path --
 * from the B1 size policy's FFT_MIN_LEN (below) up: shorter vectors are answered 0 so that the caller
 * keeps them on rustfft, which is faster than a host round trip there. */
int bdsp_hip_is_supported_fft_len_f32(int is_complex, size_t len);
int bdsp_hip_is_supported_fft_len_f64(int is_complex, size_t len);

/* The size policy of B1.  Every B1 call is a host round trip (upload, kernels, download, one synchronisation); below some
 * length the CPU code the reference runs when the plug-in declines is faster.  The reference's OpenCL backend picked its
 * device by data length for the same reason (gpu_support/ocl/mod.rs:69-80).  The trait offers two ways to decline:
 *   is_supported_fft_len -> false   sends fft() to rustfft          (time_freq/mod.rs:41-44)
 *   gpu_convolve_vector  -> None    sends convolve_signal on       (convolution.rs:504-541)
 * and this library uses them below these thresholds:
 *   FFT_MIN_LEN   is_supported_fft_len answers 0 for len (scalars) below it.  bdsp_hip_fft_* itself still transforms ANY
 *                 length it is handed.
 *   CONV_MIN_WORK gpu_convolve_vector answers 0 (None) when points x taps is below it AND the reference's next choice is
 *                 its direct form convolve_signal_scalar (real data, imp_len <= 15 scalars, or src_len <= 10 imp_len); a
 *                 complex vector the reference would send into its own overlap_discard is never declined.
 * Defaults: the crossovers measured on an MI355X box against one host core (profiles/r06_b1_crossover.txt,
 * tools/b1_crossover.py -- which a deployment can rerun on its own host).  0 = never decline.  Process-wide, thread-safe. */
#define BDSP_B1_FFT_MIN_LEN_F32 0
#define BDSP_B1_FFT_MIN_LEN_F64 1
#define BDSP_B1_CONV_MIN_WORK_F32 2
#define BDSP_B1_CONV_MIN_WORK_F64 3
size_t bdsp_hip_b1_policy_get(int key);
int bdsp_hip_b1_policy_set(int key, size_t value); /* BDSP_OK, or BDSP_ERR_UNSUPPORTED for an unknown key */

/* fn fft(is_complex, signal: &mut [T], direction)       (gpu_support/mod.rs:35)
 * In place on `len` scalars (= len/2 complex points), UNNORMALISED in both directions
 * (time_freq/mod.rs:47-58 contract).  inverse = 0 forward, 1 inverse.  Returns BDSP_OK or a
 * negative backend code (the Rust shim panics on != 0, as the OpenCL impl does). */
int bdsp_hip_fft_f32(int is_complex, float *signal, size_t len, int inverse);
int bdsp_hip_fft_f64(int is_complex, double *signal, size_t len, int inverse);

/* fn gpu_convolve_vector(is_complex, source, target, imp_resp) -> Option<Range<usize>>
 *                                                        (gpu_support/mod.rs:24-29)
 * Computes the reference's centred circular convolution (time_freq/mod.rs:455-473)
 *     y[i] = sum_k x[(i + ceil(M/2) - 1 - k) mod N] * h[k]
 * for EVERY output including the wrap-around head and tail (the OpenCL backend leaves those to
 * the CPU and never fills the tail, convolution.rs:509-527).  Returns 1 and sets
 * [*range_start, *range_end) = [0, src_len) for Some(range); returns 0 for None (declined:
 * empty input or taps longer than the signal); negative on backend failure. */
int bdsp_hip_convolve_vector_f32(int is_complex, const float *src, size_t src_len, float *dst,
                                 size_t dst_len, const float *imp, size_t imp_len,
                                 size_t *range_start, size_t *range_end);
int bdsp_hip_convolve_vector_f64(int is_complex, const double *src, size_t src_len, double *dst,
                                 size_t dst_len, const double *imp, size_t imp_len,
                                 size_t *range_start, size_t *range_end);

/* fn overlap_discard(x_time, tmp, x_freq, h_freq, imp_len, step_size) -> usize
 *                                                        (gpu_support/mod.rs:38-45)
 * All lengths in scalars (factor 2 against the caller's complex view, convolution.rs:401-412).
 * fft_len = h_len.  Blocks start at scalar position 0 and advance by step_size while
 * pos + fft_len < x_len; each block's valid part tmp[imp_len-2 .. fft_len] lands at
 * x_time[pos + imp_len/2 ..]; the LAST block's full time-domain result is left in `tmp` and the
 * final position is returned (ocl/mod.rs:426-520).  tmp[0 .. imp_len/2] (the caller's scalar
 * head, convolution.rs:376-385) is copied to x_time[0 .. imp_len/2] first, as the OpenCL impl
 * does.  The 1/fft_len scaling is applied here (h_freq arrives unscaled).  x_freq is scratch the
 * reference passes along; it is not touched.  The return value is a POSITION; failure is reported out of band:
 * the call clears the calling thread's bdsp_hip_last_error() on entry and leaves a message there (and returns 0)
 * if the backend failed -- check the message, not the position. */
size_t bdsp_hip_overlap_discard_f32(float *x_time, size_t x_len, float *tmp, size_t tmp_len,
                                    float *x_freq, size_t x_freq_len, const float *h_freq,
                                    size_t h_len, size_t imp_len, size_t step_size);
size_t bdsp_hip_overlap_discard_f64(double *x_time, size_t x_len, double *tmp, size_t tmp_len,
                                    double *x_freq, size_t x_freq_len, const double *h_freq,
                                    size_t h_len, size_t imp_len, size_t step_size);

/* ==========================================================================================
 * B2 -- device-resident vectors behind the reference's C facade names
 *       (interop/src/facade32.rs; facade64.rs is generated from it by facade64_create.pl).
 * Ownership follows the facade: operations take the handle BY VALUE (it moves in) and hand it
 * back inside the result struct; borrowed operands are const pointers.
 * ======================================================================================== */
typedef struct VecBuf32 VecBuf32; /* InteropVec<f32>, interop/src/lib.rs:16-22 */
typedef struct VecBuf64 VecBuf64;

/* #[repr(C)] VectorInteropResult<T> { result_code: i32, vector: Box<T> }  (lib.rs:203-212) */
typedef struct { int32_t result_code; VecBuf32 *vector; } VectorInteropResult32;
typedef struct { int32_t result_code; VecBuf64 *vector; } VectorInteropResult64;

/* domain: 0 = time, otherwise frequency (facade32.rs:24-28).  Window ids (lib.rs:153-164):
 * 0 triangular, 1 Hamming(0.54), 2 Blackman-Harris, 3 rectangular; 4 = Hann (generalised
 * Hamming alpha 0.5) is an ADDITION so windowed_fft(Hann) needs no host callback.
 * Conv function ids (lib.rs:166-192): 0 sinc, otherwise raised cosine(rolloff).
 * Padding option ids (lib.rs:194-200): 0 End, 1 Surround, otherwise Center. */

VecBuf32 *new32(int32_t is_complex, int32_t domain, float init_value, size_t length,
                float delta);                                  /* facade32.rs:22-41 */
VecBuf32 *new_with_performance_options32(int32_t is_complex, int32_t domain, float init_value,
                                         size_t length, float delta, size_t core_limit,
                                         int early_temp_allocation); /* :43-70 (options ignored) */
void delete_vector32(VecBuf32 *vector);                        /* facade32.rs:17-19 */
VecBuf32 *clone32(VecBuf32 *vector); /* facade32.rs:687-692; consumes its argument like the reference (Box by value) */
VecBuf32 *bdsp_hip_vec_clone32(const VecBuf32 *vector); /* non-consuming copy (addition) */
float get_value32(const VecBuf32 *vector, size_t index);     /* facade32.rs:105-107 (one-element download) */
size_t get_len32(const VecBuf32 *vector);                      /* facade32.rs:138-140 */
size_t get_points32(const VecBuf32 *vector);                   /* facade32.rs:148-150 */
float get_delta32(const VecBuf32 *vector);                     /* facade32.rs:153-155 */
int32_t is_complex32(const VecBuf32 *vector);                  /* facade32.rs:115-121 */
int32_t get_domain32(const VecBuf32 *vector);                  /* facade32.rs:130-135 */
/* data32 (facade32.rs:158-160) returns a pointer into host memory in the reference.  Here the
 * data lives in HBM: data32 downloads into a host mirror owned by the handle (valid until the
 * next call on that handle) and returns it. */
const float *data32(VecBuf32 *vector);
/* overwrite_data32 (facade32.rs:827-846): uploads `len` scalars into the front of the vector.
 * The reference rejects len >= vector.len() (strict `<`, :834); we accept len <= vector.len()
 * and return code 7 (InvalidArgumentLength) beyond that. */
VectorInteropResult32 overwrite_data32(VecBuf32 *vector, const float *data, size_t len);
void set_len32(VecBuf32 *vector, size_t len);               /* facade32.rs:143-145 */

VectorInteropResult32 real_offset32(VecBuf32 *vector, float value);          /* facade32.rs:363-365 */
VectorInteropResult32 real_scale32(VecBuf32 *vector, float value);           /* facade32.rs:368-370 */
VectorInteropResult32 complex_offset32(VecBuf32 *vector, float re, float im);/* facade32.rs:532-538 */
VectorInteropResult32 complex_scale32(VecBuf32 *vector, float re, float im); /* facade32.rs:541-547 */
VectorInteropResult32 add32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:173-175 */
VectorInteropResult32 sub32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:178-180 */
VectorInteropResult32 mul32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:188-190 */
VectorInteropResult32 div32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:183-185 */
VectorInteropResult32 conj32(VecBuf32 *vector);                              /* facade32.rs:579-581 */
VectorInteropResult32 multiply_complex_exponential32(VecBuf32 *vector, float a, float b); /* facade32.rs:695-701 */
VectorInteropResult32 magnitude32(VecBuf32 *vector);                         /* facade32.rs:559-561 */
VectorInteropResult32 magnitude_squared32(VecBuf32 *vector);                 /* facade32.rs:574-576 */
VectorInteropResult32 to_real32(VecBuf32 *vector);                           /* facade32.rs:584-586 */
VectorInteropResult32 to_imag32(VecBuf32 *vector);                           /* facade32.rs:589-591 */
VectorInteropResult32 phase32(VecBuf32 *vector);                             /* facade32.rs:662-664 */
VectorInteropResult32 to_complex32(VecBuf32 *vector);                        /* facade32.rs:418-420 */
VectorInteropResult32 reverse32(VecBuf32 *vector);                           /* facade32.rs:1142-1144 */
VectorInteropResult32 swap_halves32(VecBuf32 *vector);                       /* facade32.rs:527-529 */
VectorInteropResult32 zero_pad32(VecBuf32 *vector, size_t points, int32_t padding_option); /* facade32.rs:330-338 */
VectorInteropResult32 zero_interleave32(VecBuf32 *vector, int32_t factor);   /* facade32.rs:340-345 */
VectorInteropResult32 fft_shift32(VecBuf32 *vector);                         /* facade32.rs:963-965 */
VectorInteropResult32 ifft_shift32(VecBuf32 *vector);                        /* facade32.rs:967-969 */
VectorInteropResult32 mirror32(VecBuf32 *vector);                            /* facade32.rs:959-961 */
VectorInteropResult32 apply_window32(VecBuf32 *vector, int32_t window);      /* facade32.rs:980-983 */
VectorInteropResult32 unapply_window32(VecBuf32 *vector, int32_t window);    /* facade32.rs:987-994 */
VectorInteropResult32 plain_fft32(VecBuf32 *vector);                         /* facade32.rs:672-674 */
VectorInteropResult32 plain_ifft32(VecBuf32 *vector);                        /* facade32.rs:682-684 */
VectorInteropResult32 fft32(VecBuf32 *vector);                               /* facade32.rs:934-936 */
VectorInteropResult32 ifft32(VecBuf32 *vector);                              /* facade32.rs:944-946 */
VectorInteropResult32 windowed_fft32(VecBuf32 *vector, int32_t window);      /* facade32.rs:997-1000 */
VectorInteropResult32 windowed_ifft32(VecBuf32 *vector, int32_t window);     /* facade32.rs:1011-1014 */
VectorInteropResult32 convolve_signal32(VecBuf32 *vector, const VecBuf32 *impulse_response); /* facade32.rs:1171-1176 */
VectorInteropResult32 interpolatef32(VecBuf32 *vector, int32_t impulse_response, float rolloff,
                                     float interpolation_factor, float delay, size_t conv_len); /* :1334-1348 */

/* FFT-domain interpolation family and the symmetric (real, odd-length) transforms */
VectorInteropResult32 interpolatei32(VecBuf32 *vector, int32_t frequency_response, float rolloff,
                                     int32_t interpolation_factor);            /* facade32.rs:1426-1434 */
VectorInteropResult32 interpolate32(VecBuf32 *vector, int32_t frequency_response, float rolloff,
                                    size_t dest_points, float delay);          /* facade32.rs:1378-1387 */
VectorInteropResult32 interpft32(VecBuf32 *vector, size_t dest_points);        /* facade32.rs:1390-1395 */
VectorInteropResult32 decimatei32(VecBuf32 *vector, uint32_t decimation_factor, uint32_t delay); /* facade32.rs:1147-1153 */
VectorInteropResult32 multiply_frequency_response32(VecBuf32 *vector, int32_t frequency_response,
                                                    float rolloff, float ratio); /* facade32.rs:1293-1301 */
VectorInteropResult32 plain_sfft32(VecBuf32 *vector);                          /* facade32.rs:677-679 */
VectorInteropResult32 sfft32(VecBuf32 *vector);                                /* facade32.rs:939-941 */
VectorInteropResult32 windowed_sfft32(VecBuf32 *vector, int32_t window);       /* facade32.rs:1004-1007 */
VectorInteropResult32 plain_sifft32(VecBuf32 *vector);                         /* facade32.rs:949-951 */
VectorInteropResult32 sifft32(VecBuf32 *vector);                               /* facade32.rs:954-956 */
VectorInteropResult32 windowed_sifft32(VecBuf32 *vector, int32_t window);      /* facade32.rs:1018-1024 */

/* Correlation, function convolution, real interpolation, wrap-around binary ops, callback variants.
 * Callbacks (interop/src/lib.rs:245-377) cannot run on the device: the host samples them once into a
 * table (window: every point, or the first half mirrored when is_symmetric; convolution: 2*len+1
 * weights; frequency response: one value per bin) and a device kernel applies the table. */
typedef float (*bdsp_window_fn32)(const void *window_data, size_t n, size_t length);   /* lib.rs:306-311 */
typedef float (*bdsp_real_fn32)(const void *function_data, float x);                     /* lib.rs:245-250 */
VecBuf32 *new_with_detailed_performance_options32(int32_t is_complex, int32_t domain, float init_value,
        size_t length, float delta, size_t core_limit, size_t med_dual_core_threshold,
        size_t med_multi_core_threshold, size_t large_dual_core_threshold,
        size_t large_multi_core_threshold);                                         /* facade32.rs:70-102 (options ignored) */
void set_value32(VecBuf32 *vector, size_t index, float value);                        /* facade32.rs:110-112 */
size_t get_allocated_len32(const VecBuf32 *vector);                                 /* facade32.rs:168-170 */
const float *complex_data32(VecBuf32 *vector);  /* facade32.rs:163-165; interleaved pairs, host mirror like data32 */
VectorInteropResult32 complex_divide32(VecBuf32 *vector, float re, float im);           /* facade32.rs:550-556 */
VectorInteropResult32 add_vector32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:704-709 */
VectorInteropResult32 sub_vector32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:712-717 */
VectorInteropResult32 div_vector32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:720-725 */
VectorInteropResult32 mul_vector32(VecBuf32 *vector, const VecBuf32 *operand);      /* facade32.rs:728-733 */
VectorInteropResult32 add_smaller_vector32(VecBuf32 *vector, const VecBuf32 *operand); /* facade32.rs:736-741 */
VectorInteropResult32 sub_smaller_vector32(VecBuf32 *vector, const VecBuf32 *operand); /* facade32.rs:744-749 */
VectorInteropResult32 div_smaller_vector32(VecBuf32 *vector, const VecBuf32 *operand); /* facade32.rs:752-757 */
VectorInteropResult32 mul_smaller_vector32(VecBuf32 *vector, const VecBuf32 *operand); /* facade32.rs:760-765 */
VectorInteropResult32 prepare_argument32(VecBuf32 *vector);                         /* facade32.rs:1156-1158 */
VectorInteropResult32 prepare_argument_padded32(VecBuf32 *vector);                  /* facade32.rs:1161-1163 */
VectorInteropResult32 correlate32(VecBuf32 *vector, const VecBuf32 *other);         /* facade32.rs:1166-1168 */
VectorInteropResult32 convolve32(VecBuf32 *vector, int32_t impulse_response, float rolloff, float ratio,
                                 size_t len);                                       /* facade32.rs:1231-1240 */
VectorInteropResult32 convolve_real32(VecBuf32 *vector, bdsp_real_fn32 impulse_response,
                                      const void *impulse_response_data, bool is_symmetric, float ratio,
                                      size_t len);                                  /* facade32.rs:1183-1203 */
VectorInteropResult32 multiply_frequency_response_real32(VecBuf32 *vector, bdsp_real_fn32 frequency_response,
                                      const void *frequency_response_data, bool is_symmetric,
                                      float ratio);                                   /* facade32.rs:1247-1262 */
/* Getters into a second vector (facade32.rs:564-668).  The source handle is CONSUMED (the reference takes it
 * by value), the destination is resized to `points` reals; like the reference these return 9 on success
 * (convert_void, interop/src/lib.rs:100-105) -- a quirk kept for link compatibility. */
int32_t get_real32(VecBuf32 *vector, VecBuf32 *destination);              /* facade32.rs:652-654 */
int32_t get_imag32(VecBuf32 *vector, VecBuf32 *destination);              /* facade32.rs:657-659 */
int32_t get_magnitude32(VecBuf32 *vector, VecBuf32 *destination);         /* facade32.rs:564-566 */
int32_t get_magnitude_squared32(VecBuf32 *vector, VecBuf32 *destination); /* facade32.rs:569-571 */
int32_t get_phase32(VecBuf32 *vector, VecBuf32 *destination);             /* facade32.rs:667-669 */
VectorInteropResult32 interpolate_lin32(VecBuf32 *vector, float interpolation_factor, float delay);     /* facade32.rs:1437-1443 */
VectorInteropResult32 interpolate_hermite32(VecBuf32 *vector, float interpolation_factor, float delay); /* facade32.rs:1446-1452 */
VectorInteropResult32 apply_custom_window32(VecBuf32 *vector, bdsp_window_fn32 window, const void *window_data,
                                            bool is_symmetric);                     /* facade32.rs:1030-1044 */
VectorInteropResult32 unapply_custom_window32(VecBuf32 *vector, bdsp_window_fn32 window, const void *window_data,
                                              bool is_symmetric);                   /* facade32.rs:1049-1063 */
VectorInteropResult32 windowed_custom_fft32(VecBuf32 *vector, bdsp_window_fn32 window, const void *window_data,
                                            bool is_symmetric);                     /* facade32.rs:1068-1082 */
VectorInteropResult32 windowed_custom_sfft32(VecBuf32 *vector, bdsp_window_fn32 window, const void *window_data,
                                             bool is_symmetric);                    /* facade32.rs:1087-1101 */
VectorInteropResult32 windowed_custom_ifft32(VecBuf32 *vector, bdsp_window_fn32 window, const void *window_data,
                                             bool is_symmetric);                    /* facade32.rs:1106-1120 */
VectorInteropResult32 windowed_custom_sifft32(VecBuf32 *vector, bdsp_window_fn32 window, const void *window_data,
                                              bool is_symmetric);                   /* facade32.rs:1125-1139 */

VecBuf64 *new64(int32_t is_complex, int32_t domain, double init_value, size_t length, double delta);
VecBuf64 *new_with_performance_options64(int32_t is_complex, int32_t domain, double init_value,
                                         size_t length, double delta, size_t core_limit,
                                         int early_temp_allocation);
void delete_vector64(VecBuf64 *vector);
VecBuf64 *clone64(VecBuf64 *vector);
VecBuf64 *bdsp_hip_vec_clone64(const VecBuf64 *vector);
double get_value64(const VecBuf64 *vector, size_t index);
size_t get_len64(const VecBuf64 *vector);
size_t get_points64(const VecBuf64 *vector);
double get_delta64(const VecBuf64 *vector);
int32_t is_complex64(const VecBuf64 *vector);
int32_t get_domain64(const VecBuf64 *vector);
const double *data64(VecBuf64 *vector);
VectorInteropResult64 overwrite_data64(VecBuf64 *vector, const double *data, size_t len);
void set_len64(VecBuf64 *vector, size_t len);
VectorInteropResult64 real_offset64(VecBuf64 *vector, double value);
VectorInteropResult64 real_scale64(VecBuf64 *vector, double value);
VectorInteropResult64 complex_offset64(VecBuf64 *vector, double re, double im);
VectorInteropResult64 complex_scale64(VecBuf64 *vector, double re, double im);
VectorInteropResult64 add64(VecBuf64 *vector, const VecBuf64 *operand);
VectorInteropResult64 sub64(VecBuf64 *vector, const VecBuf64 *operand);
VectorInteropResult64 mul64(VecBuf64 *vector, const VecBuf64 *operand);
VectorInteropResult64 div64(VecBuf64 *vector, const VecBuf64 *operand);
VectorInteropResult64 conj64(VecBuf64 *vector);
VectorInteropResult64 multiply_complex_exponential64(VecBuf64 *vector, double a, double b);
VectorInteropResult64 magnitude64(VecBuf64 *vector);
VectorInteropResult64 magnitude_squared64(VecBuf64 *vector);
VectorInteropResult64 to_real64(VecBuf64 *vector);
VectorInteropResult64 to_imag64(VecBuf64 *vector);
VectorInteropResult64 phase64(VecBuf64 *vector);
VectorInteropResult64 to_complex64(VecBuf64 *vector);
VectorInteropResult64 reverse64(VecBuf64 *vector);
VectorInteropResult64 swap_halves64(VecBuf64 *vector);
VectorInteropResult64 zero_pad64(VecBuf64 *vector, size_t points, int32_t padding_option);
VectorInteropResult64 zero_interleave64(VecBuf64 *vector, int32_t factor);
VectorInteropResult64 fft_shift64(VecBuf64 *vector);
VectorInteropResult64 ifft_shift64(VecBuf64 *vector);
VectorInteropResult64 mirror64(VecBuf64 *vector);
VectorInteropResult64 apply_window64(VecBuf64 *vector, int32_t window);
VectorInteropResult64 unapply_window64(VecBuf64 *vector, int32_t window);
VectorInteropResult64 plain_fft64(VecBuf64 *vector);
VectorInteropResult64 plain_ifft64(VecBuf64 *vector);
VectorInteropResult64 fft64(VecBuf64 *vector);
VectorInteropResult64 ifft64(VecBuf64 *vector);
VectorInteropResult64 windowed_fft64(VecBuf64 *vector, int32_t window);
VectorInteropResult64 windowed_ifft64(VecBuf64 *vector, int32_t window);
VectorInteropResult64 convolve_signal64(VecBuf64 *vector, const VecBuf64 *impulse_response);
VectorInteropResult64 interpolatef64(VecBuf64 *vector, int32_t impulse_response, double rolloff,
                                     double interpolation_factor, double delay, size_t conv_len);

VectorInteropResult64 interpolatei64(VecBuf64 *vector, int32_t frequency_response, double rolloff,
                                     int32_t interpolation_factor);
VectorInteropResult64 interpolate64(VecBuf64 *vector, int32_t frequency_response, double rolloff,
                                    size_t dest_points, double delay);
VectorInteropResult64 interpft64(VecBuf64 *vector, size_t dest_points);
VectorInteropResult64 decimatei64(VecBuf64 *vector, uint32_t decimation_factor, uint32_t delay);
VectorInteropResult64 multiply_frequency_response64(VecBuf64 *vector, int32_t frequency_response,
                                                    double rolloff, double ratio);
VectorInteropResult64 plain_sfft64(VecBuf64 *vector);
VectorInteropResult64 sfft64(VecBuf64 *vector);
VectorInteropResult64 windowed_sfft64(VecBuf64 *vector, int32_t window);
VectorInteropResult64 plain_sifft64(VecBuf64 *vector);
VectorInteropResult64 sifft64(VecBuf64 *vector);
VectorInteropResult64 windowed_sifft64(VecBuf64 *vector, int32_t window);

/* Correlation, function convolution, real interpolation, wrap-around binary ops, callback variants.
 * Callbacks (interop/src/lib.rs:245-377) cannot run on the device: the host samples them once into a
 * table (window: every point, or the first half mirrored when is_symmetric; convolution: 2*len+1
 * weights; frequency response: one value per bin) and a device kernel applies the table. */
typedef double (*bdsp_window_fn64)(const void *window_data, size_t n, size_t length);   /* lib.rs:306-311 */
typedef double (*bdsp_real_fn64)(const void *function_data, double x);                     /* lib.rs:245-250 */
VecBuf64 *new_with_detailed_performance_options64(int32_t is_complex, int32_t domain, double init_value,
        size_t length, double delta, size_t core_limit, size_t med_dual_core_threshold,
        size_t med_multi_core_threshold, size_t large_dual_core_threshold,
        size_t large_multi_core_threshold);                                         /* facade32.rs:70-102 (options ignored) */
void set_value64(VecBuf64 *vector, size_t index, double value);                        /* facade32.rs:110-112 */
size_t get_allocated_len64(const VecBuf64 *vector);                                 /* facade32.rs:168-170 */
const double *complex_data64(VecBuf64 *vector);  /* facade32.rs:163-165; interleaved pairs, host mirror like data64 */
VectorInteropResult64 complex_divide64(VecBuf64 *vector, double re, double im);           /* facade32.rs:550-556 */
VectorInteropResult64 add_vector64(VecBuf64 *vector, const VecBuf64 *operand);      /* facade32.rs:704-709 */
VectorInteropResult64 sub_vector64(VecBuf64 *vector, const VecBuf64 *operand);      /* facade32.rs:712-717 */
VectorInteropResult64 div_vector64(VecBuf64 *vector, const VecBuf64 *operand);      /* facade32.rs:720-725 */
VectorInteropResult64 mul_vector64(VecBuf64 *vector, const VecBuf64 *operand);      /* facade32.rs:728-733 */
VectorInteropResult64 add_smaller_vector64(VecBuf64 *vector, const VecBuf64 *operand); /* facade32.rs:736-741 */
VectorInteropResult64 sub_smaller_vector64(VecBuf64 *vector, const VecBuf64 *operand); /* facade32.rs:744-749 */
VectorInteropResult64 div_smaller_vector64(VecBuf64 *vector, const VecBuf64 *operand); /* facade32.rs:752-757 */
VectorInteropResult64 mul_smaller_vector64(VecBuf64 *vector, const VecBuf64 *operand); /* facade32.rs:760-765 */
VectorInteropResult64 prepare_argument64(VecBuf64 *vector);                         /* facade32.rs:1156-1158 */
VectorInteropResult64 prepare_argument_padded64(VecBuf64 *vector);                  /* facade32.rs:1161-1163 */
VectorInteropResult64 correlate64(VecBuf64 *vector, const VecBuf64 *other);         /* facade32.rs:1166-1168 */
VectorInteropResult64 convolve64(VecBuf64 *vector, int32_t impulse_response, double rolloff, double ratio,
                                 size_t len);                                       /* facade32.rs:1231-1240 */
VectorInteropResult64 convolve_real64(VecBuf64 *vector, bdsp_real_fn64 impulse_response,
                                      const void *impulse_response_data, bool is_symmetric, double ratio,
                                      size_t len);                                  /* facade32.rs:1183-1203 */
VectorInteropResult64 multiply_frequency_response_real64(VecBuf64 *vector, bdsp_real_fn64 frequency_response,
                                      const void *frequency_response_data, bool is_symmetric,
                                      double ratio);                                   /* facade32.rs:1247-1262 */
/* Getters into a second vector (facade32.rs:564-668).  The source handle is CONSUMED (the reference takes it
 * by value), the destination is resized to `points` reals; like the reference these return 9 on success
 * (convert_void, interop/src/lib.rs:100-105) -- a quirk kept for link compatibility. */
int32_t get_real64(VecBuf64 *vector, VecBuf64 *destination);              /* facade32.rs:652-654 */
int32_t get_imag64(VecBuf64 *vector, VecBuf64 *destination);              /* facade32.rs:657-659 */
int32_t get_magnitude64(VecBuf64 *vector, VecBuf64 *destination);         /* facade32.rs:564-566 */
int32_t get_magnitude_squared64(VecBuf64 *vector, VecBuf64 *destination); /* facade32.rs:569-571 */
int32_t get_phase64(VecBuf64 *vector, VecBuf64 *destination);             /* facade32.rs:667-669 */
VectorInteropResult64 interpolate_lin64(VecBuf64 *vector, double interpolation_factor, double delay);     /* facade32.rs:1437-1443 */
VectorInteropResult64 interpolate_hermite64(VecBuf64 *vector, double interpolation_factor, double delay); /* facade32.rs:1446-1452 */
VectorInteropResult64 apply_custom_window64(VecBuf64 *vector, bdsp_window_fn64 window, const void *window_data,
                                            bool is_symmetric);                     /* facade32.rs:1030-1044 */
VectorInteropResult64 unapply_custom_window64(VecBuf64 *vector, bdsp_window_fn64 window, const void *window_data,
                                              bool is_symmetric);                   /* facade32.rs:1049-1063 */
VectorInteropResult64 windowed_custom_fft64(VecBuf64 *vector, bdsp_window_fn64 window, const void *window_data,
                                            bool is_symmetric);                     /* facade32.rs:1068-1082 */
VectorInteropResult64 windowed_custom_sfft64(VecBuf64 *vector, bdsp_window_fn64 window, const void *window_data,
                                             bool is_symmetric);                    /* facade32.rs:1087-1101 */
VectorInteropResult64 windowed_custom_ifft64(VecBuf64 *vector, bdsp_window_fn64 window, const void *window_data,
                                             bool is_symmetric);                    /* facade32.rs:1106-1120 */
VectorInteropResult64 windowed_custom_sifft64(VecBuf64 *vector, bdsp_window_fn64 window, const void *window_data,
                                              bool is_symmetric);                   /* facade32.rs:1125-1139 */

/* Device pointer of the handle's live buffer (valid until the next mutating call); lets the
 * batch driver feed RCCL without a host round trip.  No reference counterpart. */
void *bdsp_hip_vec_device_ptr32(VecBuf32 *vector);
void *bdsp_hip_vec_device_ptr64(VecBuf64 *vector);

/* Statistics, sums and dot products (interop/src/facade32.rs:193-327, 846-931; vector/src/vector_types/general/
 * statistics.rs, dot_products.rs).  One pass over the vector on the device (sums accumulate in double), the small
 * result comes back by value.  Minimum / maximum: first occurrence, complex values ordered by norm(); the complex
 * `rms` is sqrt(sum(z*z)/n) -- a complex number, as in the reference (statistics.rs:331, 344).  The `real_*` entry
 * points walk all interleaved scalars of a complex vector, the `complex_*` ones walk pairs.  `*_prec*` variants
 * return double-precision results (the reference uses compensated summation there). */
typedef struct { float re, im; } bdsp_complex32;   /* num_complex::Complex32, #[repr(C)] */
typedef struct { double re, im; } bdsp_complex64;
/* #[repr(C)] Statistics<T>  (statistics.rs:11-31) */
typedef struct { float sum; size_t count; float average, rms, min; size_t min_index; float max; size_t max_index; } Statistics32;
typedef struct { double sum; size_t count; double average, rms, min; size_t min_index; double max; size_t max_index; } Statistics64;
typedef struct { bdsp_complex32 sum; size_t count; bdsp_complex32 average, rms, min; size_t min_index; bdsp_complex32 max; size_t max_index; } ComplexStatistics32;
typedef struct { bdsp_complex64 sum; size_t count; bdsp_complex64 average, rms, min; size_t min_index; bdsp_complex64 max; size_t max_index; } ComplexStatistics64;
/* #[repr(C)] ScalarInteropResult<T> { result_code, result }  (interop/src/lib.rs:229-242) */
typedef struct { int32_t result_code; float result; } ScalarInteropResult32;
typedef struct { int32_t result_code; double result; } ScalarInteropResult64;
typedef struct { int32_t result_code; bdsp_complex32 result; } ComplexScalarInteropResult32;
typedef struct { int32_t result_code; bdsp_complex64 result; } ComplexScalarInteropResult64;
Statistics32 real_statistics32(const VecBuf32 *vector);                          /* facade32.rs:223-226 */
ComplexStatistics32 complex_statistics32(const VecBuf32 *vector);                /* facade32.rs:229-232 */
float real_sum32(const VecBuf32 *vector);                                          /* facade32.rs:235-237 */
float real_sum_sq32(const VecBuf32 *vector);                                       /* facade32.rs:240-242 */
bdsp_complex32 complex_sum32(const VecBuf32 *vector);                            /* facade32.rs:245-247 */
bdsp_complex32 complex_sum_sq32(const VecBuf32 *vector);                         /* facade32.rs:250-252 */
ScalarInteropResult32 real_dot_product32(const VecBuf32 *vector, const VecBuf32 *operand);             /* facade32.rs:193-201 */
ComplexScalarInteropResult32 complex_dot_product32(const VecBuf32 *vector, const VecBuf32 *operand);   /* facade32.rs:204-220 */
Statistics64 real_statistics_prec32(const VecBuf32 *vector);                     /* facade32.rs:292-295 */
ComplexStatistics64 complex_statistics_prec32(const VecBuf32 *vector);           /* facade32.rs:298-301 */
double real_sum_prec32(const VecBuf32 *vector);                                  /* facade32.rs:304-306 */
double real_sum_sq_prec32(const VecBuf32 *vector);                               /* facade32.rs:309-311 */
bdsp_complex64 complex_sum_prec32(const VecBuf32 *vector);                       /* facade32.rs:314-316 */
bdsp_complex64 complex_sum_sq_prec32(const VecBuf32 *vector);                    /* facade32.rs:319-321 */
ScalarInteropResult64 real_dot_product_prec32(const VecBuf32 *vector, const VecBuf32 *operand);            /* facade32.rs:254-270 */
ComplexScalarInteropResult64 complex_dot_product_prec32(const VecBuf32 *vector, const VecBuf32 *operand);  /* facade32.rs:273-289 */
/* statistics_split (statistics.rs:389-426): element j goes to bucket j % len with index j / len; len <= 16, else code 7 */
int32_t real_statistics_split32(const VecBuf32 *vector, Statistics32 *data, size_t len);               /* facade32.rs:848-864 */
int32_t complex_statistics_split32(const VecBuf32 *vector, ComplexStatistics32 *data, size_t len);     /* facade32.rs:867-886 */
int32_t real_statistics_split_prec32(const VecBuf32 *vector, Statistics64 *data, size_t len);          /* facade32.rs:889-907 */
int32_t complex_statistics_split_prec32(const VecBuf32 *vector, ComplexStatistics64 *data, size_t len);/* facade32.rs:910-931 */

Statistics64 real_statistics64(const VecBuf64 *vector);                          /* facade32.rs:223-226 */
ComplexStatistics64 complex_statistics64(const VecBuf64 *vector);                /* facade32.rs:229-232 */
double real_sum64(const VecBuf64 *vector);                                          /* facade32.rs:235-237 */
double real_sum_sq64(const VecBuf64 *vector);                                       /* facade32.rs:240-242 */
bdsp_complex64 complex_sum64(const VecBuf64 *vector);                            /* facade32.rs:245-247 */
bdsp_complex64 complex_sum_sq64(const VecBuf64 *vector);                         /* facade32.rs:250-252 */
ScalarInteropResult64 real_dot_product64(const VecBuf64 *vector, const VecBuf64 *operand);             /* facade32.rs:193-201 */
ComplexScalarInteropResult64 complex_dot_product64(const VecBuf64 *vector, const VecBuf64 *operand);   /* facade32.rs:204-220 */
Statistics64 real_statistics_prec64(const VecBuf64 *vector);                     /* facade32.rs:292-295 */
ComplexStatistics64 complex_statistics_prec64(const VecBuf64 *vector);           /* facade32.rs:298-301 */
double real_sum_prec64(const VecBuf64 *vector);                                  /* facade32.rs:304-306 */
double real_sum_sq_prec64(const VecBuf64 *vector);                               /* facade32.rs:309-311 */
bdsp_complex64 complex_sum_prec64(const VecBuf64 *vector);                       /* facade32.rs:314-316 */
bdsp_complex64 complex_sum_sq_prec64(const VecBuf64 *vector);                    /* facade32.rs:319-321 */
ScalarInteropResult64 real_dot_product_prec64(const VecBuf64 *vector, const VecBuf64 *operand);            /* facade32.rs:254-270 */
ComplexScalarInteropResult64 complex_dot_product_prec64(const VecBuf64 *vector, const VecBuf64 *operand);  /* facade32.rs:273-289 */
/* statistics_split (statistics.rs:389-426): element j goes to bucket j % len with index j / len; len <= 16, else code 7 */
int32_t real_statistics_split64(const VecBuf64 *vector, Statistics64 *data, size_t len);               /* facade32.rs:848-864 */
int32_t complex_statistics_split64(const VecBuf64 *vector, ComplexStatistics64 *data, size_t len);     /* facade32.rs:867-886 */
int32_t real_statistics_split_prec64(const VecBuf64 *vector, Statistics64 *data, size_t len);          /* facade32.rs:889-907 */
int32_t complex_statistics_split_prec64(const VecBuf64 *vector, ComplexStatistics64 *data, size_t len);/* facade32.rs:910-931 */

/* ----------------------------------------------------------------------------------------
 * Per-element math family, differences / running sums, phase wrapping, real<->complex pairs, split / merge,
 * host callbacks (trigonometry_and_powers.rs, real_ops.rs, diff_sum.rs, complex_to_real.rs:674-770,
 * data_reorganization.rs:484-555, mapping.rs).  TrigOps / PowerOps work on real and complex vectors (complex
 * functions as in num-complex 0.4); abs / wrap / unwrap / the *_approx family poison a complex vector.  The
 * "approximated" functions are the standard ones, as in the reference's build without explicit SIMD
 * (simd_extensions/approx_fallback.rs).  cum_sum carries the running sum in double.
 * glibc's <math.h> declares powf32 / expf32 / powf64 / expf64 itself (the _Float32 / _Float64 functions of
 * ISO/IEC TS 18661-3) and the reference's facade uses the very same names for different functions: the C
 * declarations below carry a bdsp_ prefix and bind to the facade's symbol name, so the library still exports
 * `powf32` etc. for ctypes / Rust callers and this header can be included next to <math.h>.
 * -------------------------------------------------------------------------------------- */
#define BDSP_FACADE_SYMBOL(name) __asm__(#name)
VectorInteropResult32 diff32(VecBuf32 *vector);                        /* facade32.rs:348-350 */
VectorInteropResult32 diff_with_start32(VecBuf32 *vector);             /* facade32.rs:353-355 */
VectorInteropResult32 cum_sum32(VecBuf32 *vector);                     /* facade32.rs:358-360 */
VectorInteropResult32 abs32(VecBuf32 *vector);                         /* facade32.rs:373-375 */
VectorInteropResult32 sqrt32(VecBuf32 *vector);                        /* facade32.rs:378-380 */
VectorInteropResult32 square32(VecBuf32 *vector);                      /* facade32.rs:383-385 */
VectorInteropResult32 root32(VecBuf32 *vector, float value);           /* facade32.rs:388-390 */
VectorInteropResult32 bdsp_powf32(VecBuf32 *vector, float value) BDSP_FACADE_SYMBOL(powf32);           /* facade32.rs:393-395 */
VectorInteropResult32 ln32(VecBuf32 *vector);                          /* facade32.rs:398-400 */
VectorInteropResult32 exp32(VecBuf32 *vector);                         /* facade32.rs:403-405 */
VectorInteropResult32 log32(VecBuf32 *vector, float value);            /* facade32.rs:408-410 */
VectorInteropResult32 bdsp_expf32(VecBuf32 *vector, float value) BDSP_FACADE_SYMBOL(expf32);           /* facade32.rs:413-415 */
VectorInteropResult32 sin32(VecBuf32 *vector);                         /* facade32.rs:423-425 */
VectorInteropResult32 cos32(VecBuf32 *vector);                         /* facade32.rs:428-430 */
VectorInteropResult32 tan32(VecBuf32 *vector);
VectorInteropResult32 asin32(VecBuf32 *vector);
VectorInteropResult32 acos32(VecBuf32 *vector);
VectorInteropResult32 atan32(VecBuf32 *vector);
VectorInteropResult32 sinh32(VecBuf32 *vector);
VectorInteropResult32 cosh32(VecBuf32 *vector);
VectorInteropResult32 tanh32(VecBuf32 *vector);
VectorInteropResult32 asinh32(VecBuf32 *vector);
VectorInteropResult32 acosh32(VecBuf32 *vector);
VectorInteropResult32 atanh32(VecBuf32 *vector);                       /* facade32.rs:433-479 */
VectorInteropResult32 ln_approx32(VecBuf32 *vector);                   /* facade32.rs:482-484 */
VectorInteropResult32 exp_approx32(VecBuf32 *vector);
VectorInteropResult32 sin_approx32(VecBuf32 *vector);
VectorInteropResult32 cos_approx32(VecBuf32 *vector);
VectorInteropResult32 log_approx32(VecBuf32 *vector, float value);
VectorInteropResult32 expf_approx32(VecBuf32 *vector, float value);
VectorInteropResult32 powf_approx32(VecBuf32 *vector, float value);    /* facade32.rs:487-514 */
VectorInteropResult32 wrap32(VecBuf32 *vector, float value);           /* facade32.rs:517-519 */
VectorInteropResult32 unwrap32(VecBuf32 *vector, float value);         /* facade32.rs:522-524; sequential recurrence */
/* callbacks are host code: the vector makes one host round trip, one call per element in index order */
VectorInteropResult32 map_inplace_real32(VecBuf32 *vector, float (*map)(float, size_t));                  /* facade32.rs:594-600 */
VectorInteropResult32 map_inplace_complex32(VecBuf32 *vector, bdsp_complex32 (*map)(bdsp_complex32, size_t)); /* facade32.rs:603-609 */
typedef struct { int32_t result_code; const void *result; } PointerInteropResult;  /* ScalarInteropResult<*const c_void> */
PointerInteropResult map_aggregate_real32(const VecBuf32 *vector, const void *(*map)(float, size_t),
                                          const void *(*aggregate)(const void *, const void *));       /* facade32.rs:614-629 */
PointerInteropResult map_aggregate_complex32(const VecBuf32 *vector, const void *(*map)(bdsp_complex32, size_t),
                                             const void *(*aggregate)(const void *, const void *));    /* facade32.rs:634-649 */
/* the source handle is CONSUMED, the answer is convert_void(Ok) = 9 (interop/src/lib.rs:100-105) */
int32_t get_real_imag32(VecBuf32 *vector, VecBuf32 *real, VecBuf32 *imag);                              /* facade32.rs:768-774 */
int32_t get_mag_phase32(VecBuf32 *vector, VecBuf32 *mag, VecBuf32 *phase);                              /* facade32.rs:777-783 */
VectorInteropResult32 set_real_imag32(VecBuf32 *vector, const VecBuf32 *real, const VecBuf32 *imag);    /* facade32.rs:786-792 */
VectorInteropResult32 set_mag_phase32(VecBuf32 *vector, const VecBuf32 *mag, const VecBuf32 *phase);    /* facade32.rs:795-801 */
int32_t split_into32(const VecBuf32 *vector, VecBuf32 **targets, size_t len);                           /* facade32.rs:804-811; 9 = ok */
VectorInteropResult32 merge32(VecBuf32 *vector, VecBuf32 *const *sources, size_t len);                  /* facade32.rs:814-824 */
typedef bdsp_complex32 (*bdsp_complex_fn32)(const void *function_data, float x);                        /* lib.rs:279-284 */
VectorInteropResult32 convolve_complex32(VecBuf32 *vector, bdsp_complex_fn32 impulse_response,
                                         const void *impulse_response_data, bool is_symmetric, float ratio,
                                         size_t len);                                                   /* facade32.rs:1206-1222 */
VectorInteropResult32 multiply_frequency_response_complex32(VecBuf32 *vector, bdsp_complex_fn32 frequency_response,
                                         const void *frequency_response_data, bool is_symmetric,
                                         float ratio);                                                  /* facade32.rs:1269-1284 */
VectorInteropResult32 interpolatef_custom32(VecBuf32 *vector, bdsp_real_fn32 impulse_response,
                                            const void *impulse_response_data, bool is_symmetric,
                                            float interpolation_factor, float delay, size_t len);       /* facade32.rs:1308-1326 */
VectorInteropResult32 interpolate_custom32(VecBuf32 *vector, bdsp_real_fn32 frequency_response,
                                           const void *frequency_response_data, bool is_symmetric,
                                           size_t dest_points, float delay);                            /* facade32.rs:1353-1369 */
VectorInteropResult32 interpolatei_custom32(VecBuf32 *vector, bdsp_real_fn32 frequency_response,
                                            const void *frequency_response_data, bool is_symmetric,
                                            int32_t interpolation_factor);                              /* facade32.rs:1402-1417 */

VectorInteropResult64 diff64(VecBuf64 *vector);                        /* facade32.rs:348-350 */
VectorInteropResult64 diff_with_start64(VecBuf64 *vector);             /* facade32.rs:353-355 */
VectorInteropResult64 cum_sum64(VecBuf64 *vector);                     /* facade32.rs:358-360 */
VectorInteropResult64 abs64(VecBuf64 *vector);                         /* facade32.rs:373-375 */
VectorInteropResult64 sqrt64(VecBuf64 *vector);                        /* facade32.rs:378-380 */
VectorInteropResult64 square64(VecBuf64 *vector);                      /* facade32.rs:383-385 */
VectorInteropResult64 root64(VecBuf64 *vector, double value);           /* facade32.rs:388-390 */
VectorInteropResult64 bdsp_powf64(VecBuf64 *vector, double value) BDSP_FACADE_SYMBOL(powf64);           /* facade32.rs:393-395 */
VectorInteropResult64 ln64(VecBuf64 *vector);                          /* facade32.rs:398-400 */
VectorInteropResult64 exp64(VecBuf64 *vector);                         /* facade32.rs:403-405 */
VectorInteropResult64 log64(VecBuf64 *vector, double value);            /* facade32.rs:408-410 */
VectorInteropResult64 bdsp_expf64(VecBuf64 *vector, double value) BDSP_FACADE_SYMBOL(expf64);           /* facade32.rs:413-415 */
VectorInteropResult64 sin64(VecBuf64 *vector);                         /* facade32.rs:423-425 */
VectorInteropResult64 cos64(VecBuf64 *vector);                         /* facade32.rs:428-430 */
VectorInteropResult64 tan64(VecBuf64 *vector);
VectorInteropResult64 asin64(VecBuf64 *vector);
VectorInteropResult64 acos64(VecBuf64 *vector);
VectorInteropResult64 atan64(VecBuf64 *vector);
VectorInteropResult64 sinh64(VecBuf64 *vector);
VectorInteropResult64 cosh64(VecBuf64 *vector);
VectorInteropResult64 tanh64(VecBuf64 *vector);
VectorInteropResult64 asinh64(VecBuf64 *vector);
VectorInteropResult64 acosh64(VecBuf64 *vector);
VectorInteropResult64 atanh64(VecBuf64 *vector);                       /* facade32.rs:433-479 */
VectorInteropResult64 ln_approx64(VecBuf64 *vector);                   /* facade32.rs:482-484 */
VectorInteropResult64 exp_approx64(VecBuf64 *vector);
VectorInteropResult64 sin_approx64(VecBuf64 *vector);
VectorInteropResult64 cos_approx64(VecBuf64 *vector);
VectorInteropResult64 log_approx64(VecBuf64 *vector, double value);
VectorInteropResult64 expf_approx64(VecBuf64 *vector, double value);
VectorInteropResult64 powf_approx64(VecBuf64 *vector, double value);    /* facade32.rs:487-514 */
VectorInteropResult64 wrap64(VecBuf64 *vector, double value);           /* facade32.rs:517-519 */
VectorInteropResult64 unwrap64(VecBuf64 *vector, double value);         /* facade32.rs:522-524; sequential recurrence */
/* callbacks are host code: the vector makes one host round trip, one call per element in index order */
VectorInteropResult64 map_inplace_real64(VecBuf64 *vector, double (*map)(double, size_t));                  /* facade32.rs:594-600 */
VectorInteropResult64 map_inplace_complex64(VecBuf64 *vector, bdsp_complex64 (*map)(bdsp_complex64, size_t)); /* facade32.rs:603-609 */
PointerInteropResult map_aggregate_real64(const VecBuf64 *vector, const void *(*map)(double, size_t),
                                          const void *(*aggregate)(const void *, const void *));       /* facade32.rs:614-629 */
PointerInteropResult map_aggregate_complex64(const VecBuf64 *vector, const void *(*map)(bdsp_complex64, size_t),
                                             const void *(*aggregate)(const void *, const void *));    /* facade32.rs:634-649 */
/* the source handle is CONSUMED, the answer is convert_void(Ok) = 9 (interop/src/lib.rs:100-105) */
int32_t get_real_imag64(VecBuf64 *vector, VecBuf64 *real, VecBuf64 *imag);                              /* facade32.rs:768-774 */
int32_t get_mag_phase64(VecBuf64 *vector, VecBuf64 *mag, VecBuf64 *phase);                              /* facade32.rs:777-783 */
VectorInteropResult64 set_real_imag64(VecBuf64 *vector, const VecBuf64 *real, const VecBuf64 *imag);    /* facade32.rs:786-792 */
VectorInteropResult64 set_mag_phase64(VecBuf64 *vector, const VecBuf64 *mag, const VecBuf64 *phase);    /* facade32.rs:795-801 */
int32_t split_into64(const VecBuf64 *vector, VecBuf64 **targets, size_t len);                           /* facade32.rs:804-811; 9 = ok */
VectorInteropResult64 merge64(VecBuf64 *vector, VecBuf64 *const *sources, size_t len);                  /* facade32.rs:814-824 */
typedef bdsp_complex64 (*bdsp_complex_fn64)(const void *function_data, double x);                        /* lib.rs:279-284 */
VectorInteropResult64 convolve_complex64(VecBuf64 *vector, bdsp_complex_fn64 impulse_response,
                                         const void *impulse_response_data, bool is_symmetric, double ratio,
                                         size_t len);                                                   /* facade32.rs:1206-1222 */
VectorInteropResult64 multiply_frequency_response_complex64(VecBuf64 *vector, bdsp_complex_fn64 frequency_response,
                                         const void *frequency_response_data, bool is_symmetric,
                                         double ratio);                                                  /* facade32.rs:1269-1284 */
VectorInteropResult64 interpolatef_custom64(VecBuf64 *vector, bdsp_real_fn64 impulse_response,
                                            const void *impulse_response_data, bool is_symmetric,
                                            double interpolation_factor, double delay, size_t len);       /* facade32.rs:1308-1326 */
VectorInteropResult64 interpolate_custom64(VecBuf64 *vector, bdsp_real_fn64 frequency_response,
                                           const void *frequency_response_data, bool is_symmetric,
                                           size_t dest_points, double delay);                            /* facade32.rs:1353-1369 */
VectorInteropResult64 interpolatei_custom64(VecBuf64 *vector, bdsp_real_fn64 frequency_response,
                                            const void *frequency_response_data, bool is_symmetric,
                                            int32_t interpolation_factor);                              /* facade32.rs:1402-1417 */

/* Bridges for foreign-function layers that cannot describe callbacks RETURNING structs (Python ctypes): these
 * functions have the facade's callback signatures and forward to a pointer-style callback.  Pass
 * bdsp_hip_complex_fn_bridge32 as the `bdsp_complex_fn32` and a bdsp_complex_bridge32 as its function_data; for
 * map_inplace_complex32 (no data argument) register the target for the calling thread first. */
typedef struct { void (*fn)(void *ctx, float x, float *re_im_out); void *ctx; } bdsp_complex_bridge32;
typedef struct { void (*fn)(void *ctx, double x, double *re_im_out); void *ctx; } bdsp_complex_bridge64;
bdsp_complex32 bdsp_hip_complex_fn_bridge32(const void *bridge, float x);
bdsp_complex64 bdsp_hip_complex_fn_bridge64(const void *bridge, double x);
void bdsp_hip_set_map_complex_bridge32(void (*fn)(float re, float im, size_t index, float *re_im_out));
void bdsp_hip_set_map_complex_bridge64(void (*fn)(double re, double im, size_t index, double *re_im_out));
bdsp_complex32 bdsp_hip_map_complex_bridge32(bdsp_complex32 value, size_t index);
bdsp_complex64 bdsp_hip_map_complex_bridge64(bdsp_complex64 value, size_t index);

/* ========================================================================================
 * B2m -- matrix / batch API: `rows` equally long vectors in one allocation, every operation a
 *        batched launch.  The reference's matrix crate has no C facade; these entry points mirror
 *        its Rust API (matrix/src/lib.rs:195-208 row loop, matrix/src/time_freq.rs:49-530 trait
 *        forwarding, MatrixMxN::convolve_signal with a matrix of impulse responses :439-483 ->
 *        DspVec::convolve_mat, vector/src/vector_types/time_freq/mod.rs:365-453).
 *        Operations return the facade result codes (0, -1 poisoned, 1..14, <= -100 backend).
 * ======================================================================================== */
typedef struct MatBuf32 MatBuf32;
typedef struct MatBuf64 MatBuf64;
MatBuf32 *bdsp_hip_mat_new32(int32_t is_complex, int32_t domain, size_t rows, size_t row_len, float delta); /* row_len in scalars; zero filled */
void bdsp_hip_mat_delete32(MatBuf32 *m);
size_t bdsp_hip_mat_rows32(const MatBuf32 *m);        /* col_len() of the reference (number of row vectors) */
size_t bdsp_hip_mat_row_len32(const MatBuf32 *m);     /* row_len(): scalars per row */
size_t bdsp_hip_mat_row_points32(const MatBuf32 *m);
int32_t bdsp_hip_mat_is_complex32(const MatBuf32 *m);
int32_t bdsp_hip_mat_get_domain32(const MatBuf32 *m);
float bdsp_hip_mat_get_delta32(const MatBuf32 *m);
void *bdsp_hip_mat_device_ptr32(MatBuf32 *m);         /* [rows][row_len] contiguous */
int32_t bdsp_hip_mat_upload32(MatBuf32 *m, const float *data, size_t len);   /* len == rows*row_len */
int32_t bdsp_hip_mat_download32(MatBuf32 *m, float *out, size_t len);
VecBuf32 *bdsp_hip_mat_get_row32(const MatBuf32 *m, size_t row);           /* copy of one row as a vector handle */
int32_t bdsp_hip_mat_set_row32(MatBuf32 *m, size_t row, const VecBuf32 *vector);
int32_t bdsp_hip_mat_real_scale32(MatBuf32 *m, float factor);
int32_t bdsp_hip_mat_real_offset32(MatBuf32 *m, float offset);
int32_t bdsp_hip_mat_complex_scale32(MatBuf32 *m, float re, float im);
int32_t bdsp_hip_mat_conj32(MatBuf32 *m);
int32_t bdsp_hip_mat_add32(MatBuf32 *m, const MatBuf32 *other);            /* matrix (.) matrix */
int32_t bdsp_hip_mat_sub32(MatBuf32 *m, const MatBuf32 *other);
int32_t bdsp_hip_mat_mul32(MatBuf32 *m, const MatBuf32 *other);
int32_t bdsp_hip_mat_div32(MatBuf32 *m, const MatBuf32 *other);
int32_t bdsp_hip_mat_add_vector32(MatBuf32 *m, const VecBuf32 *operand);   /* every row (.) the same vector */
int32_t bdsp_hip_mat_sub_vector32(MatBuf32 *m, const VecBuf32 *operand);
int32_t bdsp_hip_mat_mul_vector32(MatBuf32 *m, const VecBuf32 *operand);
int32_t bdsp_hip_mat_div_vector32(MatBuf32 *m, const VecBuf32 *operand);
int32_t bdsp_hip_mat_magnitude32(MatBuf32 *m);
int32_t bdsp_hip_mat_magnitude_squared32(MatBuf32 *m);
int32_t bdsp_hip_mat_to_real32(MatBuf32 *m);
int32_t bdsp_hip_mat_to_imag32(MatBuf32 *m);
int32_t bdsp_hip_mat_phase32(MatBuf32 *m);
int32_t bdsp_hip_mat_plain_fft32(MatBuf32 *m);                             /* matrix/src/time_freq.rs:53-61 */
int32_t bdsp_hip_mat_fft32(MatBuf32 *m);                                   /* :63-71 */
int32_t bdsp_hip_mat_windowed_fft32(MatBuf32 *m, int32_t window);          /* :73-81 */
int32_t bdsp_hip_mat_plain_ifft32(MatBuf32 *m);                            /* :120-128 */
int32_t bdsp_hip_mat_ifft32(MatBuf32 *m);                                  /* :130-138 */
int32_t bdsp_hip_mat_windowed_ifft32(MatBuf32 *m, int32_t window);         /* :140-148 */
int32_t bdsp_hip_mat_apply_window32(MatBuf32 *m, int32_t window);
int32_t bdsp_hip_mat_unapply_window32(MatBuf32 *m, int32_t window);
int32_t bdsp_hip_mat_swap_halves32(MatBuf32 *m);
int32_t bdsp_hip_mat_fft_shift32(MatBuf32 *m);
int32_t bdsp_hip_mat_ifft_shift32(MatBuf32 *m);
int32_t bdsp_hip_mat_zero_pad32(MatBuf32 *m, size_t points, int32_t padding_option);
int32_t bdsp_hip_mat_convolve_signal32(MatBuf32 *m, const VecBuf32 *impulse_response);      /* :421-431, one filter for all rows */
int32_t bdsp_hip_mat_convolve_signal_mat32(MatBuf32 *m, const VecBuf32 *const *impulse_responses,
                                           size_t count); /* :439-483: count == rows*rows, row-major [out][in] */
int32_t bdsp_hip_mat_interpolatef32(MatBuf32 *m, int32_t impulse_response, float rolloff,
                                    float interpolation_factor, float delay, size_t conv_len);
int32_t bdsp_hip_mat_multiply_frequency_response32(MatBuf32 *m, int32_t frequency_response, float rolloff, float ratio);

MatBuf64 *bdsp_hip_mat_new64(int32_t is_complex, int32_t domain, size_t rows, size_t row_len, double delta); /* row_len in scalars; zero filled */
void bdsp_hip_mat_delete64(MatBuf64 *m);
size_t bdsp_hip_mat_rows64(const MatBuf64 *m);        /* col_len() of the reference (number of row vectors) */
size_t bdsp_hip_mat_row_len64(const MatBuf64 *m);     /* row_len(): scalars per row */
size_t bdsp_hip_mat_row_points64(const MatBuf64 *m);
int32_t bdsp_hip_mat_is_complex64(const MatBuf64 *m);
int32_t bdsp_hip_mat_get_domain64(const MatBuf64 *m);
double bdsp_hip_mat_get_delta64(const MatBuf64 *m);
void *bdsp_hip_mat_device_ptr64(MatBuf64 *m);         /* [rows][row_len] contiguous */
int32_t bdsp_hip_mat_upload64(MatBuf64 *m, const double *data, size_t len);   /* len == rows*row_len */
int32_t bdsp_hip_mat_download64(MatBuf64 *m, double *out, size_t len);
VecBuf64 *bdsp_hip_mat_get_row64(const MatBuf64 *m, size_t row);           /* copy of one row as a vector handle */
int32_t bdsp_hip_mat_set_row64(MatBuf64 *m, size_t row, const VecBuf64 *vector);
int32_t bdsp_hip_mat_real_scale64(MatBuf64 *m, double factor);
int32_t bdsp_hip_mat_real_offset64(MatBuf64 *m, double offset);
int32_t bdsp_hip_mat_complex_scale64(MatBuf64 *m, double re, double im);
int32_t bdsp_hip_mat_conj64(MatBuf64 *m);
int32_t bdsp_hip_mat_add64(MatBuf64 *m, const MatBuf64 *other);            /* matrix (.) matrix */
int32_t bdsp_hip_mat_sub64(MatBuf64 *m, const MatBuf64 *other);
int32_t bdsp_hip_mat_mul64(MatBuf64 *m, const MatBuf64 *other);
int32_t bdsp_hip_mat_div64(MatBuf64 *m, const MatBuf64 *other);
int32_t bdsp_hip_mat_add_vector64(MatBuf64 *m, const VecBuf64 *operand);   /* every row (.) the same vector */
int32_t bdsp_hip_mat_sub_vector64(MatBuf64 *m, const VecBuf64 *operand);
int32_t bdsp_hip_mat_mul_vector64(MatBuf64 *m, const VecBuf64 *operand);
int32_t bdsp_hip_mat_div_vector64(MatBuf64 *m, const VecBuf64 *operand);
int32_t bdsp_hip_mat_magnitude64(MatBuf64 *m);
int32_t bdsp_hip_mat_magnitude_squared64(MatBuf64 *m);
int32_t bdsp_hip_mat_to_real64(MatBuf64 *m);
int32_t bdsp_hip_mat_to_imag64(MatBuf64 *m);
int32_t bdsp_hip_mat_phase64(MatBuf64 *m);
int32_t bdsp_hip_mat_plain_fft64(MatBuf64 *m);                             /* matrix/src/time_freq.rs:53-61 */
int32_t bdsp_hip_mat_fft64(MatBuf64 *m);                                   /* :63-71 */
int32_t bdsp_hip_mat_windowed_fft64(MatBuf64 *m, int32_t window);          /* :73-81 */
int32_t bdsp_hip_mat_plain_ifft64(MatBuf64 *m);                            /* :120-128 */
int32_t bdsp_hip_mat_ifft64(MatBuf64 *m);                                  /* :130-138 */
int32_t bdsp_hip_mat_windowed_ifft64(MatBuf64 *m, int32_t window);         /* :140-148 */
int32_t bdsp_hip_mat_apply_window64(MatBuf64 *m, int32_t window);
int32_t bdsp_hip_mat_unapply_window64(MatBuf64 *m, int32_t window);
int32_t bdsp_hip_mat_swap_halves64(MatBuf64 *m);
int32_t bdsp_hip_mat_fft_shift64(MatBuf64 *m);
int32_t bdsp_hip_mat_ifft_shift64(MatBuf64 *m);
int32_t bdsp_hip_mat_zero_pad64(MatBuf64 *m, size_t points, int32_t padding_option);
int32_t bdsp_hip_mat_convolve_signal64(MatBuf64 *m, const VecBuf64 *impulse_response);      /* :421-431, one filter for all rows */
int32_t bdsp_hip_mat_convolve_signal_mat64(MatBuf64 *m, const VecBuf64 *const *impulse_responses,
                                           size_t count); /* :439-483: count == rows*rows, row-major [out][in] */
int32_t bdsp_hip_mat_interpolatef64(MatBuf64 *m, int32_t impulse_response, double rolloff,
                                    double interpolation_factor, double delay, size_t conv_len);
int32_t bdsp_hip_mat_multiply_frequency_response64(MatBuf64 *m, int32_t frequency_response, double rolloff, double ratio);

/* ==========================================================================================
 * B3 -- kernels on caller-owned DEVICE memory.  `stream` is a hipStream_t passed as void*
 * (NULL = the library's own non-blocking stream; BDSP_HIP_STREAM_DEFAULT = HIP's null stream, the
 * stream a framework's "default stream" handle 0 stands for -- forward such a handle as
 * BDSP_HIP_STREAM_DEFAULT, never as 0).  Calls are asynchronous on that stream unless noted.
 * `elem` = 0 for f32, 1 for f64.  Scratch: ops that are not in place ping-pong between `data`
 * and `scratch` (same size); they return in *result_in_scratch whether the result ended in
 * `scratch` (1) or `data` (0), mirroring the reference's Buffer::trade (support_std.rs:78-82).
 * ======================================================================================== */

#define BDSP_HIP_STREAM_DEFAULT ((void *)1) /* HIP's null stream (same value as hipStreamLegacy) */

/* Unnormalised complex FFT of `batch` contiguous vectors of `points` complex each.
 * flags: BDSP_FFT_* bits. */
#define BDSP_FFT_INVERSE 1u      /* exp(+...) instead of exp(-...) */
#define BDSP_FFT_SHIFT_OUT 2u    /* fused fft_shift of the result (time_to_freq.rs:158-165) */
#define BDSP_FFT_SHIFT_IN 4u     /* fused ifft_shift of the input (freq_to_time.rs:160-168) */
#define BDSP_FFT_MAGNITUDE 8u    /* write |X| as `points` reals instead of complex (config C2) */
int bdsp_hip_dev_fft(int elem, void *data, void *scratch, size_t points, size_t batch,
                     unsigned flags, double in_scale, int window_id, double window_alpha,
                     int *result_in_scratch, void *stream);
/* Trips through device memory a power-of-two transform of `points` complex points makes (1: one workgroup-resident
 * kernel, 2 or 3: global Stockham passes) -- what bdsp_hip_dev_fft will launch for a PLAIN transform (no flags, scale or
 * window: f32 8192 points = 1, two passes once an option is fused); 0 for lengths that are not a power
 * of two (mixed-radix / chirp-z plans) or out of range.  The benchmark's per-pass figures read it from here. */
int bdsp_hip_fft_passes(int elem, size_t points);

/* Centred circular convolution (reference a9) of `batch` contiguous complex vectors of `points`
 * points with ONE shared filter of `taps` complex taps (device pointer), by fused overlap-save.
 * out must not alias in.  */
int bdsp_hip_dev_convolve(int elem, const void *in, void *out, size_t points, size_t batch,
                          const void *taps_dev, size_t taps, void *stream);

/* bdsp_hip_dev_convolve with the fused block kernel's dispatch-group shares given FOR THIS CALL: its persistent
 * workgroups are dispatched in groups (three for f32) and the groups dispatched first get a larger share of the blocks,
 * because the hardware issues the oldest wave first.  first_pct / second_pct = percent of the blocks for
 * the first / second group (defaults f32 43 / 37, f64 55 / -); (33, 33) = equal shares; (-1, -1) = the defaults, i.e.
 * bdsp_hip_dev_convolve.  The shares only move blocks between workgroups (bit-identical output); nothing outlives the
 * call and no other thread's launches see it -- the dispatch-order guard test times equal shares against the defaults
 * through this entry point.  BDSP_ERR_ARG_LENGTH for shares that leave the last group nothing to do. */
int bdsp_hip_dev_convolve_ex(int elem, const void *in, void *out, size_t points, size_t batch,
                             const void *taps_dev, size_t taps, int first_pct, int second_pct, void *stream);

/* The same convolution split in two, so a caller that reuses one filter (the batch driver, the
 * benchmark) builds its spectrum once: prepare writes bdsp_hip_conv_spectrum_points() complex
 * points (FFT of the zero-padded taps, pre-divided by the block length) to spectrum_dev;
 * convolve_prepared is the single fused overlap-save launch (taps <= 3073). */
size_t bdsp_hip_conv_spectrum_points(void);
int bdsp_hip_dev_conv_prepare(int elem, const void *taps_dev, size_t taps, void *spectrum_dev,
                              void *stream);
int bdsp_hip_dev_convolve_prepared(int elem, const void *in, void *out, size_t points, size_t batch,
                                   const void *spectrum_dev, size_t taps, void *stream);

/* Elementwise x[i] = x[i]*scale + offset is NOT offered (the reference never fuses them and a
 * fused multiply-add would round differently); separate entry points: */
int bdsp_hip_dev_real_scale(int elem, void *data, size_t len, double factor, void *stream);
int bdsp_hip_dev_real_offset(int elem, void *data, size_t len, int is_complex, double offset,
                             void *stream);

/* polyphase interpolatef (reference a13) on device memory; out holds
 * bdsp_hip_interpolatef_new_len(...) scalars. */
size_t bdsp_hip_interpolatef_new_len(int elem, size_t len, double factor);
int bdsp_hip_dev_interpolatef(int elem, const void *in, void *out, size_t len, int is_complex,
                              int function_id, double rolloff, double factor, double delay,
                              size_t conv_len, double delta, void *stream);

/* Blocks until everything queued on `stream` (NULL = library stream) has finished. */
int bdsp_hip_synchronize(void *stream);
/* Binds the calling thread's work to HIP device `ordinal` (one process per GPU: call once). */
int bdsp_hip_set_device(int ordinal);
/* Compute units of the bound device (256 on MI355X: 8 XCDs of 32); 0 without a device. */
int bdsp_hip_compute_units(void);

/* Timing hooks for bench.py: HIP events recorded on `stream`; elapsed milliseconds between
 * two recorded events.  (torch.cuda.Event only sees torch's current stream.) */
void *bdsp_hip_event_create(void);
int bdsp_hip_event_record(void *event, void *stream);
int bdsp_hip_event_elapsed_ms(void *start, void *stop, float *ms);
void bdsp_hip_event_destroy(void *event);

/* HIP graphs: capture a sequence of B2/B3 calls on one stream and replay it with a single launch
 * (small vectors are launch-bound).  Run the sequence once before capturing (tables, plans and workspace
 * are created on first use), replay on the capture stream, and keep host-facing calls (data32,
 * overwrite_data32, get_value32, B1 entry points, plain_sifft32) out of the captured region.
 * stream: hipStream_t as void*, NULL = the library's own stream (the one B2 handles use). */
int bdsp_hip_capture_begin(void *stream);
int bdsp_hip_capture_end(void *stream, void **graph_exec);
/* Drops an open capture without building a graph (a captured sequence that failed half way): the stream leaves capture
 * mode, pinned workspace and plans are released, the next capture_begin is accepted.  One capture at a time per
 * process, owned by the thread that opened it; other threads' calls neither disturb it nor are recorded by it. */
int bdsp_hip_capture_abort(void *stream);
/* The recovery route: drops the open capture from ANY thread.  For a process whose capturing thread exited or died
 * between capture_begin and capture_end -- bdsp_hip_capture_abort refuses other threads while the stream still records
 * (it accepts them once the stream no longer does), so without this every later capture_begin would be refused.  The
 * caller vouches that the owner is gone.  No-op (0) when no capture is open. */
int bdsp_hip_capture_reset(void *stream);
int bdsp_hip_graph_launch(void *graph_exec, void *stream);
void bdsp_hip_graph_destroy(void *graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* BASIC_DSP_HIP_H */
