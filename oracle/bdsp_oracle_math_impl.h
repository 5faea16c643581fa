/*
 * bdsp_oracle_math_impl.h -- TEST INFRASTRUCTURE ONLY (included by bdsp_oracle.c once per precision).
 *
 * Restatement of the reference's per-element math family, difference / running-sum operations,
 * phase wrapping and the real<->complex composition helpers:
 *   vector/src/vector_types/general/trigonometry_and_powers.rs:196-420  (TrigOps, PowerOps)
 *   vector/src/vector_types/real/real_ops.rs:236-375                     (abs, wrap, unwrap, *_approx)
 *   vector/src/vector_types/general/diff_sum.rs:65-122                   (diff, diff_with_start, cum_sum)
 *   vector/src/vector_types/complex/complex_to_real.rs:674-770           (get/set real_imag, mag_phase)
 *   vector/src/vector_types/general/data_reorganization.rs:484-555       (split_into, merge)
 * The complex functions live in the un-vendored crate num-complex ^0.4 (vector/Cargo.toml:39); what follows
 * restates its published formulas (polar forms for sqrt/powf/ln/log/expf, the logarithmic forms of the inverse
 * functions) -- finite inputs only, the crate's infinity/NaN corner cases of exp() are not reproduced.
 * The reference's "approximated" functions are, on builds without explicit SIMD, the standard functions
 * (simd_extensions/approx_fallback.rs:13-41).
 */

typedef struct { REAL re, im; } SFX(mc);

static inline SFX(mc) SFX(mc_new)(REAL re, REAL im) { SFX(mc) r = { re, im }; return r; }
static inline SFX(mc) SFX(mc_mul)(SFX(mc) a, SFX(mc) b) { return SFX(mc_new)(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline SFX(mc) SFX(mc_add)(SFX(mc) a, SFX(mc) b) { return SFX(mc_new)(a.re + b.re, a.im + b.im); }
static inline SFX(mc) SFX(mc_sub)(SFX(mc) a, SFX(mc) b) { return SFX(mc_new)(a.re - b.re, a.im - b.im); }
static inline SFX(mc) SFX(mc_div)(SFX(mc) a, SFX(mc) b)
{
    const REAL n = b.re * b.re + b.im * b.im;
    return SFX(mc_new)((a.re * b.re + a.im * b.im) / n, (a.im * b.re - a.re * b.im) / n);
}
static inline SFX(mc) SFX(mc_from_polar)(REAL r, REAL t) { return SFX(mc_new)(r * MF(cos)(t), r * MF(sin)(t)); }
static inline SFX(mc) SFX(mc_ln)(SFX(mc) z) { return SFX(mc_new)(MF(log)(MF(hypot)(z.re, z.im)), MF(atan2)(z.im, z.re)); }
static SFX(mc) SFX(mc_sqrt)(SFX(mc) z)
{
    if (z.im == 0) {
        if (!signbit(z.re)) return SFX(mc_new)(MF(sqrt)(z.re), z.im);
        const REAL im = MF(sqrt)(-z.re);
        return SFX(mc_new)(0, signbit(z.im) ? -im : im);
    }
    if (z.re == 0) {
        const REAL x = MF(sqrt)(MF(fabs)(z.im) / 2);
        return SFX(mc_new)(x, signbit(z.im) ? -x : x);
    }
    return SFX(mc_from_polar)(MF(sqrt)(MF(hypot)(z.re, z.im)), MF(atan2)(z.im, z.re) / 2);
}

#ifndef ORC_MATH_IDS
#define ORC_MATH_IDS
enum {
    ORC_M_SQRT = 0, ORC_M_SQUARE, ORC_M_POWF, ORC_M_LN, ORC_M_EXP, ORC_M_LOG, ORC_M_EXPF, ORC_M_SIN, ORC_M_COS,
    ORC_M_TAN, ORC_M_ASIN, ORC_M_ACOS, ORC_M_ATAN, ORC_M_SINH, ORC_M_COSH, ORC_M_TANH, ORC_M_ASINH, ORC_M_ACOSH,
    ORC_M_ATANH, ORC_M_ABS, ORC_M_WRAP, ORC_M_EXPF_APPROX, ORC_M_POWF_APPROX
};
#endif

static SFX(mc) SFX(mc_apply)(SFX(mc) z, int fn, REAL arg)
{
    const SFX(mc) one = { 1, 0 }, two = { 2, 0 }, i = { 0, 1 }, mi = { 0, -1 };
    switch (fn) {
    case ORC_M_SQRT: return SFX(mc_sqrt)(z);
    case ORC_M_SQUARE: return SFX(mc_mul)(z, z);
    case ORC_M_POWF:
        if (arg == 0) return one;
        return SFX(mc_from_polar)(MF(pow)(MF(hypot)(z.re, z.im), arg), MF(atan2)(z.im, z.re) * arg);
    case ORC_M_LN: return SFX(mc_ln)(z);
    case ORC_M_EXP: return SFX(mc_from_polar)(MF(exp)(z.re), z.im);
    case ORC_M_LOG: return SFX(mc_new)(MF(log)(MF(hypot)(z.re, z.im)) / MF(log)(arg), MF(atan2)(z.im, z.re) / MF(log)(arg));
    case ORC_M_EXPF: return SFX(mc_from_polar)(MF(pow)(arg, z.re), z.im * MF(log)(arg));
    case ORC_M_SIN: return SFX(mc_new)(MF(sin)(z.re) * MF(cosh)(z.im), MF(cos)(z.re) * MF(sinh)(z.im));
    case ORC_M_COS: return SFX(mc_new)(MF(cos)(z.re) * MF(cosh)(z.im), -MF(sin)(z.re) * MF(sinh)(z.im));
    case ORC_M_TAN: {
        const REAL a = z.re + z.re, b = z.im + z.im, d = MF(cos)(a) + MF(cosh)(b);
        return SFX(mc_new)(MF(sin)(a) / d, MF(sinh)(b) / d);
    }
    case ORC_M_ASIN: /* -i ln(sqrt(1 - z^2) + i z) */
        return SFX(mc_mul)(mi, SFX(mc_ln)(SFX(mc_add)(SFX(mc_sqrt)(SFX(mc_sub)(one, SFX(mc_mul)(z, z))), SFX(mc_mul)(i, z))));
    case ORC_M_ACOS: /* -i ln(i sqrt(1 - z^2) + z) */
        return SFX(mc_mul)(mi, SFX(mc_ln)(SFX(mc_add)(SFX(mc_mul)(i, SFX(mc_sqrt)(SFX(mc_sub)(one, SFX(mc_mul)(z, z)))), z)));
    case ORC_M_ATAN: /* (ln(1 + i z) - ln(1 - i z)) / (2 i) */
        if (z.re == 0 && z.im == 1) return SFX(mc_new)(0, (REAL)INFINITY);
        if (z.re == 0 && z.im == -1) return SFX(mc_new)(0, -(REAL)INFINITY);
        return SFX(mc_div)(SFX(mc_sub)(SFX(mc_ln)(SFX(mc_add)(one, SFX(mc_mul)(i, z))), SFX(mc_ln)(SFX(mc_sub)(one, SFX(mc_mul)(i, z)))),
                           SFX(mc_mul)(two, i));
    case ORC_M_SINH: return SFX(mc_new)(MF(sinh)(z.re) * MF(cos)(z.im), MF(cosh)(z.re) * MF(sin)(z.im));
    case ORC_M_COSH: return SFX(mc_new)(MF(cosh)(z.re) * MF(cos)(z.im), MF(sinh)(z.re) * MF(sin)(z.im));
    case ORC_M_TANH: {
        const REAL a = z.re + z.re, b = z.im + z.im, d = MF(cosh)(a) + MF(cos)(b);
        return SFX(mc_new)(MF(sinh)(a) / d, MF(sin)(b) / d);
    }
    case ORC_M_ASINH: /* ln(z + sqrt(1 + z^2)) */
        return SFX(mc_ln)(SFX(mc_add)(z, SFX(mc_sqrt)(SFX(mc_add)(one, SFX(mc_mul)(z, z)))));
    case ORC_M_ACOSH: /* 2 ln(sqrt((z+1)/2) + sqrt((z-1)/2)) */
        return SFX(mc_mul)(two, SFX(mc_ln)(SFX(mc_add)(SFX(mc_sqrt)(SFX(mc_div)(SFX(mc_add)(z, one), two)),
                                                         SFX(mc_sqrt)(SFX(mc_div)(SFX(mc_sub)(z, one), two)))));
    case ORC_M_ATANH: /* (ln(1 + z) - ln(1 - z)) / 2 */
        if (z.re == 1 && z.im == 0) return SFX(mc_new)((REAL)INFINITY, 0);
        if (z.re == -1 && z.im == 0) return SFX(mc_new)(-(REAL)INFINITY, 0);
        return SFX(mc_div)(SFX(mc_sub)(SFX(mc_ln)(SFX(mc_add)(one, z)), SFX(mc_ln)(SFX(mc_sub)(one, z))), two);
    default: return z;
    }
}

static REAL SFX(mr_apply)(REAL x, int fn, REAL arg)
{
    switch (fn) {
    case ORC_M_SQRT: return MF(sqrt)(x);
    case ORC_M_SQUARE: return x * x;
    case ORC_M_POWF: return MF(pow)(x, arg);
    case ORC_M_LN: return MF(log)(x);
    case ORC_M_EXP: return MF(exp)(x);
    case ORC_M_LOG: return MF(log)(x) / MF(log)(arg);          /* f32::log(base) = ln(x) / ln(base) */
    case ORC_M_EXPF: return MF(pow)(arg, x);
    case ORC_M_SIN: return MF(sin)(x);
    case ORC_M_COS: return MF(cos)(x);
    case ORC_M_TAN: return MF(tan)(x);
    case ORC_M_ASIN: return MF(asin)(x);
    case ORC_M_ACOS: return MF(acos)(x);
    case ORC_M_ATAN: return MF(atan)(x);
    case ORC_M_SINH: return MF(sinh)(x);
    case ORC_M_COSH: return MF(cosh)(x);
    case ORC_M_TANH: return MF(tanh)(x);
    case ORC_M_ASINH: return MF(asinh)(x);
    case ORC_M_ACOSH: return MF(acosh)(x);
    case ORC_M_ATANH: return MF(atanh)(x);
    case ORC_M_ABS: return MF(fabs)(x);
    case ORC_M_WRAP: return MF(fmod)(x, arg);                  /* Rust `%` on floats = fmod */
    case ORC_M_EXPF_APPROX: return MF(exp)(MF(log)(arg) * x);  /* real_ops.rs:352-362 */
    case ORC_M_POWF_APPROX: return MF(exp)(MF(log)(x) * arg);  /* real_ops.rs:364-373 */
    default: return x;
    }
}

/* every element of the vector through one function; `arg` = exponent / base / divisor where one is taken */
void SFX(orc_math)(REAL *x, size_t len, int is_complex, int fn, REAL arg)
{
    if (is_complex) {
        for (size_t i = 0; i + 1 < len; i += 2) {
            const SFX(mc) r = SFX(mc_apply)(SFX(mc_new)(x[i], x[i + 1]), fn, arg);
            x[i] = r.re; x[i + 1] = r.im;
        }
    } else {
        for (size_t i = 0; i < len; ++i) x[i] = SFX(mr_apply)(x[i], fn, arg);
    }
}

/* diff_sum.rs:65-82: returns the new length */
size_t SFX(orc_diff)(REAL *x, size_t len, int is_complex)
{
    const size_t step = is_complex ? 2 : 1;
    if (len < step) return 0;
    for (size_t j = 0; j + step < len; ++j) x[j] = x[j + step] - x[j];
    return len - step;
}

/* diff_sum.rs:84-108 */
void SFX(orc_diff_with_start)(REAL *x, size_t len, int is_complex)
{
    const size_t step = is_complex ? 2 : 1;
    for (size_t j = len; j-- > step;) x[j] = x[j] - x[j - step];
}

/* diff_sum.rs:110-121: a running sum in T */
void SFX(orc_cum_sum)(REAL *x, size_t len, int is_complex)
{
    for (size_t i = 0, j = is_complex ? 2 : 1; j < len; ++i, ++j) x[j] = x[j] + x[i];
}

/* real_ops.rs:262-284 */
void SFX(orc_unwrap)(REAL *x, size_t len, REAL divisor)
{
    const REAL half = divisor / 2;
    for (size_t i = 0, j = 1; j < len; ++i, ++j) {
        REAL diff = x[j] - x[i];
        if (diff > half) {
            diff = MF(fmod)(diff, divisor);
            diff = diff - divisor;
            x[j] = x[i] + diff;
        } else if (diff < -half) {
            diff = MF(fmod)(diff, divisor);
            diff = diff + divisor;
            x[j] = x[i] + diff;
        }
    }
}

/* complex_to_real.rs:693-712 (to_polar = (hypot, atan2)) and :749-770 (from_polar) */
void SFX(orc_get_mag_phase)(const REAL *x, size_t len, REAL *mag, REAL *phase)
{
    for (size_t i = 0; i + 1 < len; i += 2) {
        mag[i / 2] = MF(hypot)(x[i], x[i + 1]);
        phase[i / 2] = MF(atan2)(x[i + 1], x[i]);
    }
}
void SFX(orc_set_mag_phase)(const REAL *mag, const REAL *phase, size_t points, REAL *out)
{
    for (size_t p = 0; p < points; ++p) {
        out[2 * p] = mag[p] * MF(cos)(phase[p]);
        out[2 * p + 1] = mag[p] * MF(sin)(phase[p]);
    }
}

/* data_reorganization.rs:484-512: element i goes to target i % n at position i / n; targets are laid out one
 * after the other in `out` (len / n scalars each).  Returns 0, or 7 (InvalidArgumentLength). */
int SFX(orc_split_into)(const REAL *x, size_t len, int is_complex, size_t n, REAL *out)
{
    if (n == 0 || len % n != 0) return 7;
    const size_t e = is_complex ? 2 : 1, tlen = len / n;
    for (size_t i = 0; i < len / e; ++i)
        for (size_t k = 0; k < e; ++k) out[(i % n) * tlen + e * (i / n) + k] = x[e * i + k];
    return 0;
}
/* data_reorganization.rs:522-555: the inverse; `src` holds n sources of src_len scalars each */
void SFX(orc_merge)(const REAL *src, size_t src_len, int is_complex, size_t n, REAL *out)
{
    const size_t e = is_complex ? 2 : 1;
    for (size_t i = 0; i < src_len * n / e; ++i)
        for (size_t k = 0; k < e; ++k) out[e * i + k] = src[(i % n) * src_len + e * (i / n) + k];
}
