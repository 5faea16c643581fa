/*
 * bdsp_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU oracle for basic_dsp_amd: a plain-C restatement of the reference's algorithms for the hot
 * path, pinned against the reference's own known-answer tests by
 * tests/test_oracle_golden.py (fixtures under tests/golden/, transcribed by
 * tools/extract_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libbasic_dsp_hip.so) never links or calls it.
 *
 * The Rust reference cannot be built in this image (no rustc/cargo; rustfft is un-vendored), so
 * there is no oracle/_ref build; parity is pinned by golden vectors instead.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- exact-weights mode of the raised cosine (round 5) -----------------------------------------------------------------
 * The reference evaluates sin(pi x) cos(pi x beta) / (pi x) / (1 - (2 beta x)^2) in T (conv_types.rs:419-421).  Next to the
 * second singularity, |2 beta x| -> 1, numerator and denominator cancel: a tap that lands within an ulp of 1 / (2 beta)
 * without being EQUAL to it (x = -6 - 0.8 + 0.3 against 2.5) comes out of that expression without a correct digit, in the
 * reference as in its literal restatement below, and a result computed from such a tap cannot be compared with anybody
 * else's.  orc_set_exact_weights(1) makes orc_conv_time return, for the SAME argument x in T, the weight evaluated in long
 * double through the cancellation-free form cos(pi u / 2) / (1 - u^2) = sin(pi t / 2) / (t (2 - t)), t = 1 - |u|, u = 2 beta x:
 * the yardstick for the taps the reference itself cannot vouch for.  Default 0 = the literal restatement, which the
 * reference's golden vectors pin (tests/test_oracle_golden.py).  Test infrastructure; not thread-safe against a
 * concurrent switch. */
static int g_exact_weights = 0;
void orc_set_exact_weights(int on) { g_exact_weights = on; }
int orc_get_exact_weights(void) { return g_exact_weights; }
static long double orc_rc_exact(long double rolloff, long double x)
{
    const long double pi = 3.14159265358979323846264338327950288L;
    if (x == 0.0L) return 1.0L;
    const long double sinc = sinl(pi * x) / (pi * x);
    const long double u = 2.0L * rolloff * x, au = fabsl(u), t = 1.0L - au;
    long double g;
    if (fabsl(t) < 0.25L) g = t == 0.0L ? pi / 4.0L : sinl(pi * t / 2.0L) / (t * (2.0L - t));
    else g = cosl(pi * u / 2.0L) / (1.0L - u * u);
    return sinc * g;
}

/* ---- f32 instantiation ---- */
#define REAL float
#define SFX(x) x##_f32
#define R_SIN sinf
#define R_COS cosf
#define R_SQRT sqrtf
#define R_HYPOT hypotf
#define R_ATAN2 atan2f
#define R_FLOOR floorf
#define R_ROUND roundf
#define R_FABS fabsf
#define MF(name) name##f
#include "bdsp_oracle_impl.h"
#include "bdsp_oracle_math_impl.h"
#undef MF
#undef REAL
#undef SFX
#undef R_SIN
#undef R_COS
#undef R_SQRT
#undef R_HYPOT
#undef R_ATAN2
#undef R_FLOOR
#undef R_ROUND
#undef R_FABS

/* ---- f64 instantiation ---- */
#define REAL double
#define SFX(x) x##_f64
#define R_SIN sin
#define R_COS cos
#define R_SQRT sqrt
#define R_HYPOT hypot
#define R_ATAN2 atan2
#define R_FLOOR floor
#define R_ROUND round
#define R_FABS fabs
#define MF(name) name
#include "bdsp_oracle_impl.h"
#include "bdsp_oracle_math_impl.h"

/* Synthetic input generator shared by tests and bench:
 * counter-based splitmix64(seed + index) -> uniform [lo, hi). */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void orc_fill_uniform_f32(float *x, size_t len, uint64_t seed, float lo, float hi)
{
    for (size_t i = 0; i < len; ++i) {
        double u = (double)(splitmix64(seed + i) >> 11) * (1.0 / 9007199254740992.0);
        x[i] = (float)(lo + (hi - lo) * u);
    }
}

void orc_fill_uniform_f64(double *x, size_t len, uint64_t seed, double lo, double hi)
{
    for (size_t i = 0; i < len; ++i) {
        double u = (double)(splitmix64(seed + i) >> 11) * (1.0 / 9007199254740992.0);
        x[i] = lo + (hi - lo) * u;
    }
}
