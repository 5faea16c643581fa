// dsp_funcs.h -- device-side window and convolution functions, evaluated in T exactly like the
// reference evaluates them on the CPU (same formula, same operation order).
//   windows:        vector/src/window_functions.rs:26-132
//   conv functions: vector/src/conv_types.rs:391-516
#pragma once
#include "fft_core.h"

namespace bdsp {

template <typename T> __device__ __forceinline__ T dev_cos(T x);
template <> __device__ __forceinline__ float dev_cos<float>(float x) { return cosf(x); }
template <> __device__ __forceinline__ double dev_cos<double>(double x) { return cos(x); }
template <typename T> __device__ __forceinline__ T dev_sin(T x);
template <> __device__ __forceinline__ float dev_sin<float>(float x) { return sinf(x); }
template <> __device__ __forceinline__ double dev_sin<double>(double x) { return sin(x); }
template <typename T> __device__ __forceinline__ T dev_abs(T x) { return x < 0 ? -x : x; }
// cos(pi * x): the windows' angles are rational multiples of pi (2 pi n / (N - 1)), and cospi needs no
// multiplication by a rounded pi and no large-argument reduction -- closer to the exact value than
// cos(two * pi * n / (N - 1)) evaluated in T, and about half the instructions in double
template <typename T> __device__ __forceinline__ T dev_cospi(T x);
template <> __device__ __forceinline__ float dev_cospi<float>(float x) { return cospif(x); }
template <> __device__ __forceinline__ double dev_cospi<double>(double x) { return cospi(x); }
template <typename T> __device__ __forceinline__ void dev_sincospi(T x, T* s, T* c);
template <> __device__ __forceinline__ void dev_sincospi<float>(float x, float* s, float* c) { sincospif(x, s, c); }
template <> __device__ __forceinline__ void dev_sincospi<double>(double x, double* s, double* c) { sincospi(x, s, c); }

// window ids: 0 triangular, 1 Hamming(alpha), 2 Blackman-Harris, 3 rectangular
// (interop/src/lib.rs:153-164); callers map the Hann addition (id 4) to (1, alpha = 0.5).
template <typename T>
__device__ __forceinline__ T window_value(int id, T alpha, size_t n_, size_t length_)
{
    const T one = (T)1, two = (T)2;
    T n = (T)n_, length = (T)length_;
    switch (id) {
    case 0:
        return one - dev_abs((n - (length - one) / two) / (length / two));
    case 1: {
        T beta = one - alpha;
        return alpha - beta * dev_cospi(two * n / (length - one));
    }
    case 2: {
        const T four = (T)4, six = (T)6;
        const T a0 = (T)0.35875, a1 = (T)0.48829, a2 = (T)0.14128, a3 = (T)0.01168;
        return a0 - a1 * dev_cospi(two * n / (length - one)) +
               a2 * dev_cospi(four * n / (length - one)) -
               a3 * dev_cospi(six * n / (length - one));
    }
    default:
        return one;
    }
}

// Symmetric evaluation (vector_types/mod.rs:567-594): w(j) is computed for the first ceil(P/2)
// points and reused for the mirrored point P-1-j.
template <typename T>
__device__ __forceinline__ T window_value_sym(int id, T alpha, size_t i, size_t points)
{
    size_t half = points - points / 2;
    size_t j = i < half ? i : points - 1 - i;
    return window_value<T>(id, alpha, j, points);
}

// The raised cosine near its second singularity (round 5).  cos(pi beta x) / (1 - (2 beta x)^2) cancels in numerator AND
// denominator as u = |2 beta x| -> 1: evaluated as the reference writes it (conv_types.rs:419-421), a tap that lands next
// to the singularity without being EQUAL to it has no correct digit left -- in the reference too (x = -6 - 0.8 + 0.3 is 2.5
// or one ulp beside it depending on the rounding of the accumulation).  With t = 1 - u and y = (pi t / 2)^2,
//     cos(pi u / 2) / (1 - u^2) = sin(pi t / 2) / (t (2 - t)) = rc_near_num(t) / (2 - t),   rc_near_num = (pi / 2) sin(x) / x,
// a short polynomial for |t| < 0.25 (|x| < 0.4: five terms reach f32's rounding, nine f64's): no cancellation, no
// division by t, and t = 0 needs no special case.  Every raised-cosine evaluation of the library takes this form inside
// |t| < 0.25; exactly AT the singularity the reference's own value is returned (the same limit).  tests/ compares these
// taps with the oracle's exact-weights mode (oracle/: orc_set_exact_weights), everything else with its literal restatement.
template <typename T>
__device__ __forceinline__ T rc_near_num(T t)
{
    const T x = (T)1.57079632679489661923 * t, y = x * x;
    T p;
    if (sizeof(T) == 4)
        p = (T)1 + y * ((T)(-1.0 / 6) + y * ((T)(1.0 / 120) + y * ((T)(-1.0 / 5040) + y * (T)(1.0 / 362880))));
    else
        p = (T)1 + y * ((T)(-1.0 / 6) + y * ((T)(1.0 / 120) + y * ((T)(-1.0 / 5040) + y * ((T)(1.0 / 362880) + y * ((T)(-1.0 / 39916800) +
            y * ((T)(1.0 / 6227020800.0) + y * ((T)(-1.0 / 1307674368000.0) + y * (T)(1.0 / 355687428096000.0))))))));
    return (T)1.57079632679489661923 * p;
}

// conv function ids (interop/src/lib.rs:166-192): 0 sinc, otherwise raised cosine(rolloff)
template <typename T>
__device__ __forceinline__ T conv_time_value(int id, T rolloff, T x)
{
    const T one = (T)1, two = (T)2, pi = (T)3.14159265358979323846;
    if (x == (T)0) return one;
    if (id == 0) {
        T pi_x = pi * x;
        return dev_sin(pi_x) / pi_x;
    }
    const T four = two * two;
    if (dev_abs(x) == one / (two * rolloff)) {
        T arg = pi / two / rolloff;
        return dev_sin(arg) / arg * pi / four;
    }
    T pi_x = pi * x;
    T arg = two * rolloff * x;
    const T t = one - dev_abs(arg);
    if (dev_abs(t) < (T)0.25) return dev_sin(pi_x) * rc_near_num<T>(t) / pi_x / (two - t);
    return dev_sin(pi_x) * dev_cos(pi_x * rolloff) / pi_x / (one - (arg * arg));
}

// frequency-domain forms (conv_types.rs:434-449, 498-505) and the axis mapping of a spectrum in
// fft-shifted or natural order (time_freq/mod.rs:67-77)
template <typename T>
__device__ __forceinline__ T conv_freq_value(int id, T rolloff, T x)
{
    const T one = (T)1, two = (T)2, pi = (T)3.14159265358979323846;
    T ax = dev_abs(x);
    if (id == 0) return ax <= one ? one : (T)0;
    if (ax <= (one - rolloff)) return one;
    if (((one - rolloff) < ax) && (ax <= (one + rolloff)))
        return one / two * (one + dev_cos(pi / rolloff * (ax - (one - rolloff)) / two));
    return (T)0;
}

template <typename T>
__device__ __forceinline__ T fft_swap_x(bool is_fft_shifted, T x_value, T x_max)
{
    if (!is_fft_shifted) return x_value / x_max;
    if (x_value <= (T)0) return (T)1 + x_value / x_max;
    return -(x_max - x_value + (T)1) / x_max;
}

} // namespace bdsp
