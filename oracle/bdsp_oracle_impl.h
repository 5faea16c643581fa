/*
 * bdsp_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Scalar CPU restatement of the basic_dsp hot path, written from the
 * reference's behaviour, one instantiation per precision.  Included twice by bdsp_oracle.c with
 *   REAL   = float | double          (element type T of the reference)
 *   SFX(x) = x##_f32 | x##_f64       (symbol suffix)
 *   R_SIN/R_COS/R_SQRT/R_HYPOT/R_ATAN2/R_FLOOR/R_ROUND/R_FABS  libm entry points of that precision
 *
 * Arithmetic is done in REAL exactly where the reference does it in T (elementwise ops, windows,
 * tap evaluation, convolution sums), so the f32 instantiation has the reference's own rounding
 * class and the f64 instantiation (fed with up-cast inputs) is the ground truth the 1e-6 parity
 * tolerance is measured against.  FFT twiddles are produced in double and rounded once.
 *
 * All citations are relative to /root/reference.
 * Layout: complex data is interleaved [re0, im0, re1, im1, ...] (vector/src/lib.rs:272-279);
 * `len` counts scalars, `points` counts complex (or real) elements.
 */

typedef struct { REAL re, im; } SFX(cpx);

static inline SFX(cpx) SFX(cmul)(SFX(cpx) a, SFX(cpx) b)
{
    /* num-complex Mul: (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re) */
    SFX(cpx) r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}

/* ------------------------------------------------------------------------------------------
 * a2: ScaleOps / OffsetOps  (vector/src/vector_types/general/elementary.rs:283-360)
 * ---------------------------------------------------------------------------------------- */

/* scale(f): x[i] *= f over ALL interleaved scalars (elementary.rs:327-342). */
void SFX(orc_real_scale)(REAL *x, size_t len, REAL f)
{
    for (size_t i = 0; i < len; ++i) x[i] = x[i] * f;
}

/* offset(f): real vector x[i] += f; complex vector adds (f, 0) to every complex
 * (elementary.rs:283-307: Complex::new(offset, 0) added per element). */
void SFX(orc_real_offset)(REAL *x, size_t len, int is_complex, REAL f)
{
    if (is_complex) {
        for (size_t i = 0; i + 1 < len; i += 2) { x[i] = x[i] + f; x[i + 1] = x[i + 1] + (REAL)0; }
    } else {
        for (size_t i = 0; i < len; ++i) x[i] = x[i] + f;
    }
}

/* complex scale z *= c (elementary.rs:344-360) */
void SFX(orc_complex_scale)(REAL *x, size_t len, REAL re, REAL im)
{
    SFX(cpx) c = { re, im };
    SFX(cpx) *z = (SFX(cpx) *)x;
    for (size_t i = 0; i < len / 2; ++i) z[i] = SFX(cmul)(z[i], c);
}

/* complex offset z += c (elementary.rs:309-325) */
void SFX(orc_complex_offset)(REAL *x, size_t len, REAL re, REAL im)
{
    for (size_t i = 0; i + 1 < len; i += 2) { x[i] = x[i] + re; x[i + 1] = x[i + 1] + im; }
}

/* ------------------------------------------------------------------------------------------
 * a16: ElementaryOps add/sub/mul/div (elementary.rs:540-589); op: 0 add 1 sub 2 mul 3 div.
 * Complex mul/div follow num-complex (div: (a*conj(b)) / |b|^2 component-wise).
 * Returns 0 or the reference's ErrorReason code (interop/src/lib.rs:125-142): 1 = size mismatch.
 * ---------------------------------------------------------------------------------------- */
int SFX(orc_binary)(REAL *x, size_t len, const REAL *y, size_t ylen, int is_complex, int op)
{
    if (len != ylen) return 1;
    if (!is_complex || op < 2) {
        for (size_t i = 0; i < len; ++i) {
            switch (op) {
            case 0: x[i] = x[i] + y[i]; break;
            case 1: x[i] = x[i] - y[i]; break;
            case 2: x[i] = x[i] * y[i]; break;
            default: x[i] = x[i] / y[i]; break;
            }
        }
        return 0;
    }
    SFX(cpx) *a = (SFX(cpx) *)x;
    const SFX(cpx) *b = (const SFX(cpx) *)y;
    for (size_t i = 0; i < len / 2; ++i) {
        if (op == 2) {
            a[i] = SFX(cmul)(a[i], b[i]);
        } else {
            REAL n = b[i].re * b[i].re + b[i].im * b[i].im;
            SFX(cpx) r;
            r.re = (a[i].re * b[i].re + a[i].im * b[i].im) / n;
            r.im = (a[i].im * b[i].re - a[i].re * b[i].im) / n;
            a[i] = r;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a15: ComplexOps (vector/src/vector_types/complex/complex_ops.rs:81-116)
 * ---------------------------------------------------------------------------------------- */

/* multiply_complex_exponential(a, b): z[k] *= exp(j*(a*delta*k + b*delta)), evaluated by the
 * reference's running product (complex_ops.rs:95-102; single chunk = default 1-thread setting). */
void SFX(orc_multiply_complex_exponential)(REAL *x, size_t len, REAL a, REAL b, REAL delta)
{
    a = a * delta;
    b = b * delta;
    SFX(cpx) e0 = { R_COS(b), R_SIN(b) };
    SFX(cpx) e1 = { R_COS(a * (REAL)0), R_SIN(a * (REAL)0) };
    SFX(cpx) e = SFX(cmul)(e0, e1);
    SFX(cpx) inc = { R_COS(a), R_SIN(a) };
    SFX(cpx) *z = (SFX(cpx) *)x;
    for (size_t i = 0; i < len / 2; ++i) {
        z[i] = SFX(cmul)(z[i], e);
        e = SFX(cmul)(e, inc);
    }
}

/* conj (complex_ops.rs:107-116) */
void SFX(orc_conj)(REAL *x, size_t len)
{
    for (size_t i = 1; i < len; i += 2) x[i] = -x[i];
}

/* ------------------------------------------------------------------------------------------
 * a8: complex -> real (vector/src/vector_types/complex/complex_to_real.rs:374-478)
 * kind: 0 magnitude (hypot, :374-379) 1 magnitude_squared 2 to_real 3 to_imag 4 phase (atan2)
 * in: len scalars (complex); out: len/2 scalars.  out may alias x (in-place compaction,
 * vector_types/mod.rs:437-452).
 * ---------------------------------------------------------------------------------------- */
void SFX(orc_complex_to_real)(const REAL *x, size_t len, REAL *out, int kind)
{
    for (size_t i = 0; i < len / 2; ++i) {
        REAL re = x[2 * i], im = x[2 * i + 1], r;
        switch (kind) {
        case 0: r = R_HYPOT(re, im); break;
        case 1: r = re * re + im * im; break;
        case 2: r = re; break;
        case 3: r = im; break;
        default: r = R_ATAN2(im, re); break;
        }
        out[i] = r;
    }
}

/* ------------------------------------------------------------------------------------------
 * a7: windows (vector/src/window_functions.rs:26-132)
 * id: 0 triangular, 1 generalized Hamming(alpha) [alpha=0.54 default, 0.5 = Hann],
 *     2 Blackman-Harris, 3 rectangular  (ids as interop/src/lib.rs:153-164)
 * ---------------------------------------------------------------------------------------- */
REAL SFX(orc_window_value)(int id, REAL alpha, size_t n_, size_t length_)
{
    const REAL one = (REAL)1, two = (REAL)2, pi = (REAL)M_PI;
    REAL n = (REAL)n_, length = (REAL)length_;
    switch (id) {
    case 0: /* window_functions.rs:36-42 */
        return one - R_FABS((n - (length - one) / two) / (length / two));
    case 1: { /* :80-87 */
        REAL beta = one - alpha;
        return alpha - beta * R_COS(two * pi * n / (length - one));
    }
    case 2: { /* :100-115 */
        const REAL four = (REAL)4, six = (REAL)6;
        const REAL a0 = (REAL)0.35875, a1 = (REAL)0.48829, a2 = (REAL)0.14128, a3 = (REAL)0.01168;
        return a0 - a1 * R_COS(two * pi * n / (length - one))
             + a2 * R_COS(four * pi * n / (length - one))
             - a3 * R_COS(six * pi * n / (length - one));
    }
    default:
        return one;
    }
}

/* apply_window / unapply_window (time_freq/time.rs:32-66 -> multiply_window_priv
 * vector_types/mod.rs:526-597).  All four built-in windows report is_symmetric() = true, so the
 * reference evaluates w(j) for the first ceil(P/2) points and applies the SAME value to x[j] and
 * x[P-1-j] (mod.rs:567-594). */
void SFX(orc_apply_window)(REAL *x, size_t len, int is_complex, int id, REAL alpha, int unapply)
{
    size_t points = is_complex ? len / 2 : len;
    size_t half = points - points / 2; /* first half gets the middle element when P is odd */
    for (size_t j = 0; j < half; ++j) {
        REAL w = SFX(orc_window_value)(id, alpha, j, points);
        if (unapply) w = (REAL)1 / w;
        size_t m = points - 1 - j;
        if (is_complex) {
            /* complex * Complex::new(w, 0): (re*w - im*0, re*0 + im*w) */
            REAL re = x[2 * j], im = x[2 * j + 1];
            x[2 * j] = re * w - im * (REAL)0;
            x[2 * j + 1] = re * (REAL)0 + im * w;
            if (m != j) {
                re = x[2 * m]; im = x[2 * m + 1];
                x[2 * m] = re * w - im * (REAL)0;
                x[2 * m + 1] = re * (REAL)0 + im * w;
            }
        } else {
            x[j] = x[j] * w;
            if (m != j) x[m] = x[m] * w;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a18: conv functions (vector/src/conv_types.rs:391-516). id 0 = sinc, 1 = raised cosine.
 * ---------------------------------------------------------------------------------------- */
REAL SFX(orc_conv_time)(int id, REAL rolloff, REAL x)
{
    const REAL one = (REAL)1, two = (REAL)2, pi = (REAL)M_PI;
    if (x == (REAL)0) return one;
    if (id == 0) { /* conv_types.rs:476-488 */
        REAL pi_x = pi * x;
        return R_SIN(pi_x) / pi_x;
    }
    if (g_exact_weights) return (REAL)orc_rc_exact((long double)rolloff, (long double)x); /* (bdsp_oracle.c: the yardstick mode) */
    /* conv_types.rs:406-424 */
    const REAL four = two * two;
    if (R_FABS(x) == one / (two * rolloff)) {
        REAL arg = pi / two / rolloff;
        return R_SIN(arg) / arg * pi / four;
    }
    REAL pi_x = pi * x;
    REAL arg = two * rolloff * x;
    return R_SIN(pi_x) * R_COS(pi_x * rolloff) / pi_x / (one - (arg * arg));
}

REAL SFX(orc_conv_freq)(int id, REAL rolloff, REAL x)
{
    const REAL one = (REAL)1, two = (REAL)2, pi = (REAL)M_PI;
    REAL ax = R_FABS(x);
    if (id == 0) return ax <= one ? one : (REAL)0; /* conv_types.rs:498-505 */
    /* conv_types.rs:434-449 */
    if (ax <= (one - rolloff)) return one;
    if (((one - rolloff) < ax) && (ax <= (one + rolloff)))
        return one / two * (one + R_COS(pi / rolloff * (ax - (one - rolloff)) / two));
    return (REAL)0;
}

/* ------------------------------------------------------------------------------------------
 * a6: swap_halves / fft_shift / ifft_shift (vector_types/mod.rs:171-191, 509-524; freq.rs:85-91)
 * Works on `points` elements of `elem` scalars each (elem = 2 for complex).
 * ---------------------------------------------------------------------------------------- */
void SFX(orc_swap_halves)(REAL *x, size_t len, int is_complex, int forward)
{
    size_t elem = is_complex ? 2 : 1;
    size_t n = len / elem;
    if (n == 0) return;
    if (n % 2 == 0) {
        size_t h = n / 2;
        for (size_t i = 0; i < h; ++i)
            for (size_t e = 0; e < elem; ++e) {
                REAL t = x[i * elem + e];
                x[i * elem + e] = x[(i + h) * elem + e];
                x[(i + h) * elem + e] = t;
            }
    } else {
        /* cycle walk, mod.rs:181-189 */
        size_t step = forward ? n / 2 : n / 2 + 1;
        REAL temp[2] = { x[0], elem == 2 ? x[1] : (REAL)0 };
        size_t pos = step;
        for (size_t k = 0; k < n; ++k) {
            size_t pos_new = (pos + step) % n;
            for (size_t e = 0; e < elem; ++e) {
                REAL t = temp[e];
                temp[e] = x[pos * elem + e];
                x[pos * elem + e] = t;
            }
            pos = pos_new;
        }
    }
}

/* reverse (data_reorganization.rs:237-247) */
void SFX(orc_reverse)(REAL *x, size_t len, int is_complex)
{
    size_t elem = is_complex ? 2 : 1;
    size_t n = len / elem;
    for (size_t i = 0; i < n / 2; ++i)
        for (size_t e = 0; e < elem; ++e) {
            REAL t = x[i * elem + e];
            x[i * elem + e] = x[(n - 1 - i) * elem + e];
            x[(n - 1 - i) * elem + e] = t;
        }
}

/* ------------------------------------------------------------------------------------------
 * a17: zero_pad / zero_pad_b / zero_interleave (data_reorganization.rs:310-479)
 * option: 0 End, 1 Surround, 2 Center (ids interop/src/lib.rs:194-200).
 * buffered = 0: in-place variant (:310-360; Surround left = diff - (diff-1)/2);
 * buffered = 1: `_b` variant (:407-463; Surround right = diff/2).
 * out has points*elem scalars.  Returns 0 or 7 (InvalidArgumentLength).
 * Note: zero_pad_b's Center branch zeroes `left..len-len_before` (:454), leaving part of the
 * borrowed buffer uninitialised; the oracle zeroes the whole gap left..len-right like the in-place
 * variant (documented deviation: we restate the intended semantics, not the garbage).
 * ---------------------------------------------------------------------------------------- */
int SFX(orc_zero_pad)(const REAL *x, size_t len_before, int is_complex, size_t points, int option,
                      int buffered, REAL *out)
{
    size_t step = is_complex ? 2 : 1;
    size_t len = points * step;
    if (len <= len_before) return 7;
    memset(out, 0, len * sizeof(REAL));
    if (option == 0) {
        memcpy(out, x, len_before * sizeof(REAL));
    } else if (option == 1) {
        size_t diff = (len - len_before) / step;
        size_t right = buffered ? diff / 2 : (diff - 1) / 2;
        size_t left = (diff - right) * step;
        memcpy(out + left, x, len_before * sizeof(REAL));
    } else {
        size_t points_before = len_before / step;
        size_t right = (points_before / 2) * step;
        size_t left = (points_before - points_before / 2) * step;
        memcpy(out, x, left * sizeof(REAL));
        memcpy(out + len - right, x + len_before - right, right * sizeof(REAL));
    }
    return 0;
}

/* zero_interleave: out[i*factor] = in[i], zero elsewhere (data_reorganization.rs:362-401,254-302) */
void SFX(orc_zero_interleave)(const REAL *x, size_t len, int is_complex, size_t factor, REAL *out)
{
    size_t elem = is_complex ? 2 : 1;
    size_t n = len / elem;
    memset(out, 0, len * factor * sizeof(REAL));
    for (size_t i = 0; i < n; ++i)
        for (size_t e = 0; e < elem; ++e) out[i * factor * elem + e] = x[i * elem + e];
}

/* mirror (time_freq/freq.rs:52-83): [z0, z1..zP-1] -> [z0, z1..zP-1, conj(zP-1)..conj(z1)];
 * out has 2*len-2 scalars. */
void SFX(orc_mirror)(const REAL *x, size_t len, REAL *out)
{
    size_t p = len / 2;
    memcpy(out, x, len * sizeof(REAL));
    for (size_t k = 1; k < p; ++k) {
        out[2 * (p - 1 + k)] = x[2 * (p - k)];
        out[2 * (p - 1 + k) + 1] = -x[2 * (p - k) + 1];
    }
}

/* ------------------------------------------------------------------------------------------
 * a3: fft core (time_freq/mod.rs:32-63).  Unnormalised DFT, Forward = exp(-2 pi i nk/N),
 * Inverse = exp(+...), no 1/N, any N.  The arithmetic itself lives in the un-vendored crate
 * rustfft ^6 (vector/Cargo.toml:40); this is a restatement of the published definition:
 * iterative radix-2 for powers of two, Bluestein's chirp-z for every other N, plus a naive
 * O(N^2) DFT used by the tests to cross-check both.
 * ---------------------------------------------------------------------------------------- */
static void SFX(fft_pow2)(SFX(cpx) *x, size_t n, int inverse)
{
    if (n < 2) return;
    /* bit reversal */
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { SFX(cpx) t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    /* twiddle table w[k] = exp(-/+ 2 pi i k / n), k < n/2, from double */
    SFX(cpx) *w = (SFX(cpx) *)malloc(sizeof(SFX(cpx)) * (n / 2));
    for (size_t k = 0; k < n / 2; ++k) {
        double a = (inverse ? 2.0 : -2.0) * M_PI * (double)k / (double)n;
        w[k].re = (REAL)cos(a);
        w[k].im = (REAL)sin(a);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1, stride = n / len;
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < half; ++k) {
                SFX(cpx) u = x[i + k];
                SFX(cpx) v = SFX(cmul)(x[i + k + half], w[k * stride]);
                x[i + k].re = u.re + v.re;
                x[i + k].im = u.im + v.im;
                x[i + k + half].re = u.re - v.re;
                x[i + k + half].im = u.im - v.im;
            }
        }
    }
    free(w);
}

void SFX(orc_dft_naive)(const REAL *in, REAL *out, size_t points, int inverse)
{
    for (size_t k = 0; k < points; ++k) {
        double sr = 0.0, si = 0.0;
        for (size_t n = 0; n < points; ++n) {
            /* reduce n*k mod N before scaling to keep the angle accurate */
            size_t m = (size_t)(((unsigned __int128)n * k) % points);
            double a = (inverse ? 2.0 : -2.0) * M_PI * (double)m / (double)points;
            double c = cos(a), s = sin(a);
            sr += (double)in[2 * n] * c - (double)in[2 * n + 1] * s;
            si += (double)in[2 * n] * s + (double)in[2 * n + 1] * c;
        }
        out[2 * k] = (REAL)sr;
        out[2 * k + 1] = (REAL)si;
    }
}

void SFX(orc_fft)(REAL *data, size_t points, int inverse)
{
    SFX(cpx) *x = (SFX(cpx) *)data;
    size_t n = points;
    if (n < 2) return;
    if ((n & (n - 1)) == 0) { SFX(fft_pow2)(x, n, inverse); return; }
    /* Bluestein: X[k] = conj(c[k]) * sum_n (x[n] conj(c[n])) c[k-n], c[n] = exp(+/- i pi n^2 / N) */
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    SFX(cpx) *a = (SFX(cpx) *)calloc(m, sizeof(SFX(cpx)));
    SFX(cpx) *b = (SFX(cpx) *)calloc(m, sizeof(SFX(cpx)));
    SFX(cpx) *c = (SFX(cpx) *)malloc(n * sizeof(SFX(cpx)));
    for (size_t i = 0; i < n; ++i) {
        size_t sq = (size_t)(((unsigned __int128)i * i) % (2 * n));
        double ang = (inverse ? -1.0 : 1.0) * M_PI * (double)sq / (double)n;
        c[i].re = (REAL)cos(ang); /* c = exp(+i pi n^2/N) for forward */
        c[i].im = (REAL)sin(ang);
    }
    for (size_t i = 0; i < n; ++i) {
        SFX(cpx) cc = { c[i].re, -c[i].im };
        a[i] = SFX(cmul)(x[i], cc);
        b[i] = c[i];
        if (i) b[m - i] = c[i];
    }
    SFX(fft_pow2)(a, m, 0);
    SFX(fft_pow2)(b, m, 0);
    for (size_t i = 0; i < m; ++i) a[i] = SFX(cmul)(a[i], b[i]);
    SFX(fft_pow2)(a, m, 1);
    REAL inv_m = (REAL)1 / (REAL)m;
    for (size_t k = 0; k < n; ++k) {
        SFX(cpx) cc = { c[k].re, -c[k].im };
        SFX(cpx) v = { a[k].re * inv_m, a[k].im * inv_m };
        x[k] = SFX(cmul)(v, cc);
    }
    free(a); free(b); free(c);
}

/* ------------------------------------------------------------------------------------------
 * a9/a11: centred circular convolution (time_freq/mod.rs:275-361, convolve_iteration :455-473,
 * ReverseWrappingIterator :788-848):
 *     y[i] = sum_{k=0}^{M'-1} x[(i + c - 1 - k) mod N] * h'[k]
 * with (h', M', c) = (h, M, M - M/2) when M <= N, else the centre taps h[M/2 - N/2 .. M/2 + N/2)
 * and c = N/2 (mod.rs:284-296).  Sum order and precision as the reference (k ascending, in T).
 * ---------------------------------------------------------------------------------------- */
static inline size_t SFX(wrap)(long long pos, size_t n)
{
    long long m = pos % (long long)n;
    if (m < 0) m += (long long)n;
    return (size_t)m;
}

static void SFX(conv_setup)(size_t points, size_t other_points, size_t *start, size_t *count,
                            long long *conv_len)
{
    if (other_points > points) {
        size_t center = other_points / 2, cl = points / 2;
        *start = center - cl;
        *count = 2 * cl;      /* zip() stops at the shorter of full_conv_len and the tap slice */
        if (*count > points) *count = points;
        *conv_len = (long long)cl;
    } else {
        *start = 0;
        *count = other_points;
        *conv_len = (long long)(other_points - other_points / 2);
    }
}

void SFX(orc_convolve_direct_range)(const REAL *x, size_t len, const REAL *h, size_t hlen,
                                    int is_complex, REAL *out, size_t first, size_t count_out)
{
    size_t elem = is_complex ? 2 : 1;
    size_t n = len / elem, m = hlen / elem, start, count;
    long long c;
    SFX(conv_setup)(n, m, &start, &count, &c);
    const REAL *taps = h + start * elem;
    for (size_t i = first; i < first + count_out && i < n; ++i) {
        /* ReverseWrappingIterator::new(data, i + conv_len, ..) pre-decrements */
        size_t pos = SFX(wrap)((long long)i + c, n);
        if (is_complex) {
            SFX(cpx) sum = { 0, 0 };
            const SFX(cpx) *xd = (const SFX(cpx) *)x;
            const SFX(cpx) *hd = (const SFX(cpx) *)taps;
            for (size_t k = 0; k < count; ++k) {
                pos = pos > 0 ? pos - 1 : n - 1;
                SFX(cpx) p = SFX(cmul)(xd[pos], hd[k]);
                sum.re = sum.re + p.re;
                sum.im = sum.im + p.im;
            }
            out[2 * i] = sum.re;
            out[2 * i + 1] = sum.im;
        } else {
            REAL sum = 0;
            for (size_t k = 0; k < count; ++k) {
                pos = pos > 0 ? pos - 1 : n - 1;
                sum = sum + x[pos] * taps[k];
            }
            out[i] = sum;
        }
    }
}

void SFX(orc_convolve_direct)(const REAL *x, size_t len, const REAL *h, size_t hlen,
                              int is_complex, REAL *out)
{
    SFX(orc_convolve_direct_range)(x, len, h, hlen, is_complex, out, 0, len);
}

/* ------------------------------------------------------------------------------------------
 * a10: overlap_discard (time_freq/convolution.rs:292-462), complex only.  Follows the reference
 * schedule literally: fft_len = max(arg, next_pow2(4*(M-1))) (:326-331), step = fft_len-(M-1),
 * (1) scalar head of M/2 outputs (:376-385), (2) scalar tail of remainder_len/2 outputs with
 * remainder_len = x_len - x_len % fft_len (:341,387-397), block loop while pos+fft_len < x_len
 * (:413-451), store of the last block (:454-455) and of the tail (:458).
 * `in_place` semantics: x is overwritten like the reference's signal_time.
 * If fair != 0 the O(N*M) scalar tail is skipped and ALL outputs come from modular overlap-save
 * blocks instead (the "CPU-fair" baseline of BASELINE.md section 3; not the reference's schedule).
 * Returns 0, or 3 if !complex (InputMustBeComplex).
 * ---------------------------------------------------------------------------------------- */
size_t SFX(orc_next_power_of_two)(size_t value)
{
    /* convolution.rs:270-282 */
    size_t count = 0, n = value;
    if (n != 0 && (n & (n - 1)) == 0) return n;
    while (n != 0) { n >>= 1; count++; }
    return (size_t)1 << count;
}

int SFX(orc_overlap_discard)(REAL *x, size_t len, const REAL *h, size_t hlen, size_t fft_len_arg,
                             int fair)
{
    size_t x_len = len / 2, imp_len = hlen / 2;
    if (imp_len == 0 || x_len == 0) return 7;
    size_t overlap = imp_len - 1;
    size_t min_fft_len = SFX(orc_next_power_of_two)(4 * overlap);
    size_t fft_len = fft_len_arg > min_fft_len ? fft_len_arg : min_fft_len;
    if (fft_len < 2) fft_len = 2;
    size_t step = fft_len - overlap;
    SFX(cpx) *sig = (SFX(cpx) *)x;
    const SFX(cpx) *taps = (const SFX(cpx) *)h;
    SFX(cpx) *H = (SFX(cpx) *)calloc(fft_len, sizeof(SFX(cpx)));
    SFX(cpx) *blk = (SFX(cpx) *)malloc(fft_len * sizeof(SFX(cpx)));
    memcpy(H, taps, imp_len * sizeof(SFX(cpx)));
    SFX(orc_fft)((REAL *)H, fft_len, 0);
    REAL scaling = (REAL)fft_len;

    if (fair) {
        /* every output from a block; loads wrap modulo x_len; out-of-place into a copy */
        SFX(cpx) *src = (SFX(cpx) *)malloc(x_len * sizeof(SFX(cpx)));
        memcpy(src, sig, x_len * sizeof(SFX(cpx)));
        long long back = (long long)(imp_len - (imp_len - imp_len / 2)); /* M - c = M/2 */
        for (size_t o = 0; o < x_len; o += step) {
            for (size_t n = 0; n < fft_len; ++n)
                blk[n] = src[SFX(wrap)((long long)o - back + (long long)n, x_len)];
            SFX(orc_fft)((REAL *)blk, fft_len, 0);
            for (size_t n = 0; n < fft_len; ++n) {
                SFX(cpx) p = SFX(cmul)(blk[n], H[n]);
                blk[n].re = p.re / scaling;
                blk[n].im = p.im / scaling;
            }
            SFX(orc_fft)((REAL *)blk, fft_len, 1);
            for (size_t m = 0; m < step && o + m < x_len; ++m) sig[o + m] = blk[m + overlap];
        }
        free(src); free(H); free(blk);
        return 0;
    }

    size_t remainder_len = x_len - x_len % fft_len;
    SFX(cpx) *head = (SFX(cpx) *)malloc((imp_len / 2 + 1) * sizeof(SFX(cpx)));
    SFX(cpx) *end = (SFX(cpx) *)malloc((remainder_len / 2 + 1) * sizeof(SFX(cpx)));
    /* (1) and (2): scalar convolution with conv_len = (imp_len+1)/2 on the ORIGINAL signal */
    {
        REAL *tmpout = (REAL *)malloc(len * sizeof(REAL));
        SFX(orc_convolve_direct_range)(x, len, h, hlen, 1, tmpout, 0, imp_len / 2);
        memcpy(head, tmpout, (imp_len / 2) * sizeof(SFX(cpx)));
        size_t tail_first = x_len - remainder_len / 2; /* == signal_time.len() - end.len()... */
        /* reference: position = signal_time.len() - end.len() where end has remainder_len/2
         * complex (array_to_complex of remainder_len scalars), :387-388 */
        SFX(orc_convolve_direct_range)(x, len, h, hlen, 1, tmpout, tail_first, remainder_len / 2);
        memcpy(end, (SFX(cpx) *)tmpout + tail_first, (remainder_len / 2) * sizeof(SFX(cpx)));
        free(tmpout);
    }

    SFX(cpx) *tmp = (SFX(cpx) *)calloc(fft_len, sizeof(SFX(cpx)));
    SFX(cpx) *ovl = (SFX(cpx) *)malloc((overlap + 1) * sizeof(SFX(cpx)));
    size_t position = 0;
    int have_block = 0;
    if (x_len >= fft_len) {
        /* (3) first iteration */
        memcpy(ovl, sig + position + step, overlap * sizeof(SFX(cpx)));
        memcpy(blk, sig + position, fft_len * sizeof(SFX(cpx)));
        SFX(orc_fft)((REAL *)blk, fft_len, 0);
        memcpy(sig, head, (imp_len / 2) * sizeof(SFX(cpx)));
        for (size_t n = 0; n < fft_len; ++n) {
            SFX(cpx) p = SFX(cmul)(blk[n], H[n]);
            tmp[n].re = p.re / scaling;
            tmp[n].im = p.im / scaling;
        }
        SFX(orc_fft)((REAL *)tmp, fft_len, 1);
        position += step;
        have_block = 1;
        while (position + fft_len < x_len) {
            /* restore the overlap the previous store clobbered, remember the next one */
            memcpy(sig + position, ovl, overlap * sizeof(SFX(cpx)));
            memcpy(ovl, sig + position + step, overlap * sizeof(SFX(cpx)));
            memcpy(blk, sig + position, fft_len * sizeof(SFX(cpx)));
            SFX(orc_fft)((REAL *)blk, fft_len, 0);
            /* (4) store the previous block's valid part */
            memcpy(sig + position - step + imp_len / 2, tmp + imp_len - 1, step * sizeof(SFX(cpx)));
            for (size_t n = 0; n < fft_len; ++n) {
                SFX(cpx) p = SFX(cmul)(blk[n], H[n]);
                tmp[n].re = p.re / scaling;
                tmp[n].im = p.im / scaling;
            }
            SFX(orc_fft)((REAL *)tmp, fft_len, 1);
            position += step;
        }
    }
    /* (5) last block, (6) tail */
    if (have_block)
        memcpy(sig + position - step + imp_len / 2, tmp + imp_len - 1, step * sizeof(SFX(cpx)));
    memcpy(sig + x_len - remainder_len / 2, end, (remainder_len / 2) * sizeof(SFX(cpx)));
    free(H); free(blk); free(head); free(end); free(tmp); free(ovl);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Multi-threaded CPU baseline (bench.py's cpu_baseline leg only): the tail-free overlap-save above with its
 * independent blocks spread over `threads` OpenMP threads, and a radix-2 transform whose butterflies of one
 * stage are spread likewise.  The reference runs both on one thread unless the vector's MultiCoreSettings say
 * otherwise (threading.rs:210-231: `parallel()` = half the logical cores for Large operations); this is what
 * "all host cores" can do with the same algorithm.  Same arithmetic as the single-threaded functions.
 * ---------------------------------------------------------------------------------------- */
void SFX(orc_fft_pow2_mt)(REAL *data, size_t points, int inverse, int threads)
{
    SFX(cpx) *x = (SFX(cpx) *)data;
    size_t n = points;
    if (n < 2 || (n & (n - 1)) != 0) return;
    if (threads < 1) threads = 1;
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { SFX(cpx) t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    SFX(cpx) *w = (SFX(cpx) *)malloc(sizeof(SFX(cpx)) * (n / 2));
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long k = 0; k < (long long)(n / 2); ++k) {
        double a = (inverse ? 2.0 : -2.0) * M_PI * (double)k / (double)n;
        w[k].re = (REAL)cos(a);
        w[k].im = (REAL)sin(a);
    }
    /* stages that stay inside one of `parts` contiguous chunks: every thread runs all of them on its own chunks
     * (cache-resident); the last log2(parts) stages span chunks and are spread butterfly by butterfly */
    size_t parts = 1;
    while (parts * 2 <= (size_t)threads && parts * 2 <= n / 2) parts *= 2;
    size_t chunk = n / parts;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long c = 0; c < (long long)parts; ++c) {
        SFX(cpx) *xc = x + (size_t)c * chunk;
        for (size_t len = 2; len <= chunk; len <<= 1) {
            size_t half = len >> 1, stride = n / len;
            for (size_t i = 0; i < chunk; i += len)
                for (size_t k = 0; k < half; ++k) {
                    SFX(cpx) u = xc[i + k];
                    SFX(cpx) v = SFX(cmul)(xc[i + k + half], w[k * stride]);
                    xc[i + k].re = u.re + v.re;
                    xc[i + k].im = u.im + v.im;
                    xc[i + k + half].re = u.re - v.re;
                    xc[i + k + half].im = u.im - v.im;
                }
        }
    }
    for (size_t len = 2 * chunk; len <= n; len <<= 1) {
        size_t half = len >> 1, stride = n / len;
#pragma omp parallel for num_threads(threads) schedule(static)
        for (long long j = 0; j < (long long)(n / 2); ++j) {
            size_t i = ((size_t)j / half) * len, k = (size_t)j % half;
            SFX(cpx) u = x[i + k];
            SFX(cpx) v = SFX(cmul)(x[i + k + half], w[k * stride]);
            x[i + k].re = u.re + v.re;
            x[i + k].im = u.im + v.im;
            x[i + k + half].re = u.re - v.re;
            x[i + k + half].im = u.im - v.im;
        }
    }
    free(w);
}

int SFX(orc_overlap_save_mt)(REAL *x, size_t len, const REAL *h, size_t hlen, size_t fft_len_arg, int threads)
{
    size_t x_len = len / 2, imp_len = hlen / 2;
    if (imp_len == 0 || x_len == 0) return 7;
    if (threads < 1) threads = 1;
    size_t overlap = imp_len - 1;
    size_t min_fft_len = SFX(orc_next_power_of_two)(4 * overlap);
    size_t fft_len = fft_len_arg > min_fft_len ? fft_len_arg : min_fft_len;
    if (fft_len < 2) fft_len = 2;
    size_t step = fft_len - overlap;
    SFX(cpx) *sig = (SFX(cpx) *)x;
    SFX(cpx) *H = (SFX(cpx) *)calloc(fft_len, sizeof(SFX(cpx)));
    memcpy(H, h, imp_len * sizeof(SFX(cpx)));
    SFX(orc_fft)((REAL *)H, fft_len, 0);
    REAL scaling = (REAL)fft_len;
    SFX(cpx) *src = (SFX(cpx) *)malloc(x_len * sizeof(SFX(cpx)));
    memcpy(src, sig, x_len * sizeof(SFX(cpx)));
    long long back = (long long)(imp_len / 2);
    long long nblocks = (long long)((x_len + step - 1) / step);
#pragma omp parallel num_threads(threads)
    {
        SFX(cpx) *blk = (SFX(cpx) *)malloc(fft_len * sizeof(SFX(cpx)));
#pragma omp for schedule(dynamic, 4)
        for (long long b = 0; b < nblocks; ++b) {
            size_t o = (size_t)b * step;
            for (size_t n = 0; n < fft_len; ++n)
                blk[n] = src[SFX(wrap)((long long)o - back + (long long)n, x_len)];
            SFX(orc_fft)((REAL *)blk, fft_len, 0);
            for (size_t n = 0; n < fft_len; ++n) {
                SFX(cpx) p = SFX(cmul)(blk[n], H[n]);
                blk[n].re = p.re / scaling;
                blk[n].im = p.im / scaling;
            }
            SFX(orc_fft)((REAL *)blk, fft_len, 1);
            for (size_t m = 0; m < step && o + m < x_len; ++m) sig[o + m] = blk[m + overlap];
        }
        free(blk);
    }
    free(src); free(H);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a9: convolve_signal dispatcher (time_freq/convolution.rs:464-543).  Every branch computes the
 * same a9 sum; what differs is the schedule (and therefore rounding).  has_gpu = 0 here (this is
 * the CPU path).  `out` receives the result (len scalars).
 * path_out: 1 simd (:499-502, same arithmetic as direct, register-blocked) 3 overlap_discard
 *           (:530-538) 4 scalar (:540).
 * Returns 0 or an ErrorReason code (7: points < imp points).
 * ---------------------------------------------------------------------------------------- */
int SFX(orc_convolve_signal)(const REAL *x, size_t len, const REAL *h, size_t hlen, int is_complex,
                             REAL *out, int *path_out)
{
    size_t elem = is_complex ? 2 : 1;
    if (len / elem < hlen / elem) return 7;
    int path = 4;
    if (len > 1000 && hlen <= 202 && hlen > 11) path = 1;
    else if (len > 10000 && hlen > 15 && len > 10 * hlen && is_complex) path = 3;
    if (path_out) *path_out = path;
    if (path == 3) {
        memcpy(out, x, len * sizeof(REAL));
        return SFX(orc_overlap_discard)(out, len, h, hlen, SFX(orc_next_power_of_two)(hlen), 0);
    }
    SFX(orc_convolve_direct)(x, len, h, hlen, is_complex, out);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a13: interpolatef (time_freq/interpolation.rs:387-482)
 * ---------------------------------------------------------------------------------------- */
size_t SFX(orc_interpolatef_new_len)(size_t len, REAL factor)
{
    /* :406-410  new_len = round(len * factor), made even */
    size_t new_len = (size_t)R_ROUND((REAL)len * factor);
    return new_len + new_len % 2;
}

/* WrappingIterator::new(slice, pos, n) (mod.rs:725-786) yields slice[(pos+1) mod N], ... */
static void SFX(interp_scalar)(REAL *out, size_t new_points, const REAL *x, size_t points,
                               size_t elem, int fid, REAL rolloff, REAL factor, REAL delay,
                               size_t conv_len)
{
    /* interpolate_priv_scalar, interpolation.rs:92-131 */
    for (size_t i = 0; i < new_points; ++i) {
        REAL center = (REAL)(long long)i / factor;
        REAL rounded = R_FLOOR(center);
        long long start = (long long)rounded - (long long)conv_len - 1;
        REAL j = -(REAL)conv_len - (center - rounded) + delay;
        REAL sr = 0, si = 0;
        size_t pos = SFX(wrap)(start, points);
        for (size_t k = 0; k < 2 * conv_len + 1; ++k) {
            pos = pos + 1 < points ? pos + 1 : 0;
            REAL w = SFX(orc_conv_time)(fid, rolloff, j);
            if (elem == 2) {
                /* c * TT::from(w) = c * Complex(w, 0) */
                REAL re = x[2 * pos], im = x[2 * pos + 1];
                sr = sr + (re * w - im * (REAL)0);
                si = si + (re * (REAL)0 + im * w);
            } else {
                sr = sr + x[pos] * w;
            }
            j = j + (REAL)1;
        }
        if (elem == 2) { out[2 * i] = sr; out[2 * i + 1] = si; }
        else out[i] = sr;
    }
}

/* function_to_vector (interpolation.rs:159-181): taps[m] = f(-(L-1) + delay + m - offset) */
static void SFX(interp_taps)(REAL *taps, int fid, REAL rolloff, size_t conv_len, REAL offset,
                             REAL delay)
{
    REAL j = -((REAL)conv_len - (REAL)1) + delay;
    for (size_t m = 0; m < 2 * conv_len + 1; ++m) {
        taps[m] = SFX(orc_conv_time)(fid, rolloff, j - offset);
        j = j + (REAL)1;
    }
}

static void SFX(interp_simd)(REAL *out, size_t new_points, const REAL *x, size_t points,
                             size_t elem, int fid, REAL rolloff, size_t factor, REAL delay,
                             size_t conv_len)
{
    /* interpolate_priv_simd, interpolation.rs:191-290.  Per-phase tap vectors
     * (function_to_vectors :133-157: offset = shift / factor), edges by
     * interpolate_priv_simd_step (:293-315), inner region by the reversed register dot product
     * (:249-273) which reduces to  y[i] = sum_m x[c + L - 1 - m] * taps_{(f - i%f)%f}[m],
     * c = ceil(i/f)  (no wrap-around: the inner region starts (2L+1)*f outputs in). */
    size_t ntaps = 2 * conv_len + 1;
    REAL *vec = (REAL *)malloc(sizeof(REAL) * ntaps * factor);
    for (size_t s = 0; s < factor; ++s)
        SFX(interp_taps)(vec + s * ntaps, fid, rolloff, conv_len, (REAL)s / (REAL)factor, delay);
    size_t scalar_len = ntaps * factor;
    for (size_t i = 0; i < new_points; ++i) {
        REAL sr = 0, si = 0;
        int edge = (i < scalar_len) || (i + scalar_len >= new_points);
        if (new_points < 2 * scalar_len) edge = 1;
        if (edge) {
            size_t rounded = i / factor;
            const REAL *t = vec + (i % factor) * ntaps;
            size_t pos = SFX(wrap)((long long)rounded - (long long)conv_len, points);
            for (size_t m = 0; m < ntaps; ++m) {
                pos = pos + 1 < points ? pos + 1 : 0;
                if (elem == 2) {
                    REAL re = x[2 * pos], im = x[2 * pos + 1];
                    sr = sr + (re * t[m] - im * (REAL)0);
                    si = si + (re * (REAL)0 + im * t[m]);
                } else sr = sr + x[pos] * t[m];
            }
        } else {
            size_t rounded = (i + factor - 1) / factor;
            size_t end = rounded + conv_len; /* exclusive */
            size_t shift = (factor - i % factor) % factor;
            const REAL *t = vec + shift * ntaps;
            /* lowest address first, as the register loop runs (taps reversed) */
            for (size_t m = ntaps; m-- > 0;) {
                size_t n = end - 1 - m;
                if (elem == 2) {
                    sr = sr + x[2 * n] * t[m];
                    si = si + x[2 * n + 1] * t[m];
                } else sr = sr + x[n] * t[m];
            }
        }
        if (elem == 2) { out[2 * i] = sr; out[2 * i + 1] = si; }
        else out[i] = sr;
    }
    free(vec);
}

/* Dispatcher, interpolation.rs:387-482.  delay is divided by delta (:397), conv_len clamped to
 * points/2 (:399-404).  path_out: 1 = simd path (:411-445), 0 = scalar path.
 * out must hold orc_interpolatef_new_len(len, factor) scalars. */
void SFX(orc_interpolatef)(const REAL *x, size_t len, int is_complex, int fid, REAL rolloff,
                           REAL factor, REAL delay, size_t conv_len, REAL delta, REAL *out,
                           int *path_out)
{
    size_t elem = is_complex ? 2 : 1;
    size_t points = len / elem;
    delay = delay / delta;
    if (conv_len > points / 2) conv_len = points / 2;
    size_t new_len = SFX(orc_interpolatef_new_len)(len, factor);
    int simd = conv_len <= 202 && new_len >= 2000 &&
               R_FABS(R_ROUND(factor) - factor) < (REAL)1e-6;
    if (path_out) *path_out = simd;
    if (simd)
        SFX(interp_simd)(out, new_len / elem, x, points, elem, fid, rolloff,
                         (size_t)R_ROUND(factor), delay, conv_len);
    else
        SFX(interp_scalar)(out, new_len / elem, x, points, elem, fid, rolloff, factor, delay,
                           conv_len);
}

/* ------------------------------------------------------------------------------------------
 * a14: FFT-domain interpolation family (time_freq/interpolation.rs:319-376, 484-633) and
 * multiply_function_priv (time_freq/mod.rs:612-723), fft_swap_x (mod.rs:67-77).
 * ---------------------------------------------------------------------------------------- */
static REAL SFX(fft_swap_x)(int is_fft_shifted, REAL x_value, REAL x_max)
{
    if (!is_fft_shifted) return x_value / x_max;
    if (x_value <= (REAL)0) return (REAL)1 + x_value / x_max;
    return -(x_max - x_value + (REAL)1) / x_max;
}

/* multiply_function_priv for a SYMMETRIC function (all built-in responses are): the function is
 * evaluated on the negative half of the axis, j = -center + i, and the same value multiplies the
 * mirrored element (mod.rs:655-721) => element i uses j = -|i - center|,
 * center = max = (points - points%2)/2.  value *= ratio * f(fft_swap_x(j, max) * ratio). */
void SFX(orc_multiply_frequency_response)(REAL *x, size_t len, int is_complex, int fid, REAL rolloff,
                                          REAL ratio, int is_fft_shifted)
{
    size_t elem = is_complex ? 2 : 1, points = len / elem;
    size_t offset = points % 2;
    REAL maxv = (REAL)(points - offset) / (REAL)2;
    for (size_t i = 0; i < points; ++i) {
        REAL j = -maxv + (REAL)i;
        if (j > (REAL)0) j = -j; /* mirrored element reuses the value of its negative-axis partner */
        REAL f = SFX(orc_conv_freq)(fid, rolloff, SFX(fft_swap_x)(is_fft_shifted, j, maxv) * ratio);
        REAL arg = ratio * f;
        if (is_complex) {
            /* Complex * Complex::new(arg, 0) */
            REAL re = x[2 * i], im = x[2 * i + 1];
            x[2 * i] = re * arg - im * (REAL)0;
            x[2 * i + 1] = re * (REAL)0 + im * arg;
        } else {
            x[i] = x[i] * arg;
        }
    }
}

/* apply_linear_phase (interpolation.rs:319-339): two running-product complex exponentials */
static void SFX(apply_linear_phase)(REAL *x, size_t len, REAL delay)
{
    const REAL pi = (REAL)M_PI, two = (REAL)2;
    size_t points = len / 2, pos_points = points / 2, neg_points = points - pos_points;
    REAL phase_inc = two * pi * delay / (REAL)points;
    REAL start = -(REAL)neg_points * phase_inc;
    SFX(orc_multiply_complex_exponential)(x + 2 * pos_points, len - 2 * pos_points, phase_inc, start, (REAL)1);
    SFX(orc_multiply_complex_exponential)(x, 2 * pos_points, phase_inc, (REAL)0, (REAL)1);
}

/* interpolatei (interpolation.rs:484-532).  out holds len*factor scalars.  Returns 0 or 10. */
int SFX(orc_interpolatei)(const REAL *x, size_t len, int is_complex, int fid, REAL rolloff,
                          unsigned factor, REAL *out)
{
    if (factor <= 1) { memcpy(out, x, len * sizeof(REAL)); return 0; }
    size_t points = is_complex ? len / 2 : len;
    size_t np = points * factor;
    REAL *c = (REAL *)calloc(2 * np, sizeof(REAL));
    for (size_t i = 0; i < points; ++i) {
        c[2 * i * factor] = is_complex ? x[2 * i] : x[i];
        c[2 * i * factor + 1] = is_complex ? x[2 * i + 1] : (REAL)0;
    }
    SFX(orc_fft)(c, np, 0);
    SFX(orc_multiply_frequency_response)(c, 2 * np, 1, fid, rolloff, (REAL)factor, 1);
    SFX(orc_fft)(c, np, 1);
    SFX(orc_real_scale)(c, 2 * np, (REAL)1 / (REAL)np);
    if (is_complex) memcpy(out, c, 2 * np * sizeof(REAL));
    else for (size_t i = 0; i < np; ++i) out[i] = c[2 * i];
    free(c);
    return 0;
}

/* interpolate / interpft (interpolation.rs:534-605).  fid < 0: no frequency response (interpft).
 * out holds dest_points*(is_complex?2:1) scalars; *delta_out = delta / (dest_points/points). */
int SFX(orc_interpolate)(const REAL *x, size_t len, int is_complex, int fid, REAL rolloff,
                         size_t dest_points, REAL delay, REAL delta, REAL *out, REAL *delta_out)
{
    size_t points = is_complex ? len / 2 : len;
    size_t dest_len = is_complex ? 2 * dest_points : dest_points;
    REAL factorf = (REAL)dest_points / (REAL)points;
    size_t maxp = points > dest_points ? points : dest_points;
    REAL *c = (REAL *)calloc(2 * maxp, sizeof(REAL));
    for (size_t i = 0; i < points; ++i) {
        c[2 * i] = is_complex ? x[2 * i] : x[i];
        c[2 * i + 1] = is_complex ? x[2 * i + 1] : (REAL)0;
    }
    SFX(orc_fft)(c, points, 0);
    if (delay != (REAL)0) SFX(apply_linear_phase)(c, 2 * points, delay / delta);
    if (dest_len > len) {
        REAL *p = (REAL *)malloc(2 * dest_points * sizeof(REAL));
        SFX(orc_zero_pad)(c, 2 * points, 1, dest_points, 2, 0, p);
        memcpy(c, p, 2 * dest_points * sizeof(REAL));
        free(p);
        if (fid < 0) SFX(orc_real_scale)(c, 2 * dest_points, factorf);
        else SFX(orc_multiply_frequency_response)(c, 2 * dest_points, 1, fid, rolloff, factorf, 1);
    } else if (dest_len < len) {
        /* interpolate_downsample (:362-376) */
        size_t orig_len = 2 * points, neg_points = dest_points / 2, pos_points = dest_points - neg_points;
        memmove(c + 2 * pos_points, c + orig_len - 2 * neg_points, 2 * neg_points * sizeof(REAL));
        SFX(orc_real_scale)(c, 2 * dest_points, (REAL)(2 * dest_points) / (REAL)orig_len);
    }
    SFX(orc_fft)(c, dest_points, 1);
    SFX(orc_real_scale)(c, 2 * dest_points, (REAL)1 / (REAL)dest_points);
    if (delta_out) *delta_out = delta / factorf;
    if (is_complex) memcpy(out, c, 2 * dest_points * sizeof(REAL));
    else for (size_t i = 0; i < dest_points; ++i) out[i] = c[2 * i];
    free(c);
    return 0;
}

/* decimatei (interpolation.rs:607-633): out[j] = in[delay + j*factor]; returns new len (scalars) */
size_t SFX(orc_decimatei)(const REAL *x, size_t len, int is_complex, unsigned factor, unsigned delay, REAL *out)
{
    size_t elem = is_complex ? 2 : 1, points = len / elem, j = 0;
    for (size_t i = delay; i < points; i += factor, ++j)
        for (size_t e = 0; e < elem; ++e) out[j * elem + e] = x[i * elem + e];
    return j * elem;
}

/* ------------------------------------------------------------------------------------------
 * convolve(function, ratio, len) in the time domain: convolve_function_priv
 * (time_freq/mod.rs:174-213).  WrappingIterator pre-increments (mod.rs:741-757), so the window of
 * output i is x[i-L .. i+L] (wrapping), weighted by f(-j*ratio), j = -L .. L accumulated in REAL.
 * conv_len is clamped to the number of points (:197).
 * ---------------------------------------------------------------------------------------- */
void SFX(orc_convolve_function)(const REAL *x, size_t len, int is_complex, int fid, REAL rolloff,
                                REAL ratio, size_t conv_len, REAL *out)
{
    size_t elem = is_complex ? 2 : 1, points = len / elem;
    if (points == 0) return;
    if (conv_len > points) conv_len = points;
    for (size_t i = 0; i < points; ++i) {
        REAL sre = 0, sim = 0;
        REAL j = -(REAL)conv_len;
        for (size_t k = 0; k < 2 * conv_len + 1; ++k) {
            size_t p = SFX(wrap)((long long)i - (long long)conv_len + (long long)k, points);
            REAL w = SFX(orc_conv_time)(fid, rolloff, -j * ratio);
            if (is_complex) {
                /* Complex * Complex::new(w, 0) */
                REAL re = x[2 * p], im = x[2 * p + 1];
                sre = sre + (re * w - im * (REAL)0);
                sim = sim + (re * (REAL)0 + im * w);
            } else {
                sre = sre + x[p] * w;
            }
            j = j + (REAL)1;
        }
        out[i * elem] = sre;
        if (is_complex) out[i * elem + 1] = sim;
    }
}

/* ------------------------------------------------------------------------------------------
 * Cross correlation (time_freq/correlation.rs:96-160).
 * prepare_argument(_padded): [zero_pad_b(2*points-1, Surround)] -> plain_fft -> conj.
 * correlate: zero_pad_b(other.points, Surround) -> plain_fft -> mul(other) -> plain_ifft ->
 * scale(1/points) -> swap_halves.  All vectors complex.  `arg` is the PREPARED argument
 * (arg_len scalars); out holds arg_len scalars.  Returns 0, 7 (other not longer than self:
 * zero_pad_b fails) -- the caller checks domains.
 * ---------------------------------------------------------------------------------------- */
int SFX(orc_prepare_argument)(const REAL *x, size_t len, int padded, REAL *out)
{
    size_t points = len / 2, np = padded ? 2 * points - 1 : points;
    if (padded) {
        if (SFX(orc_zero_pad)(x, len, 1, np, 1, 1, out)) return 7;
    } else {
        memcpy(out, x, len * sizeof(REAL));
    }
    SFX(orc_fft)(out, np, 0);
    SFX(orc_conj)(out, 2 * np);
    return 0;
}

int SFX(orc_correlate)(const REAL *x, size_t len, const REAL *arg, size_t arg_len, REAL *out)
{
    size_t points = arg_len / 2;
    int code = SFX(orc_zero_pad)(x, len, 1, points, 1, 1, out);
    if (code) return code;
    SFX(orc_fft)(out, points, 0);
    SFX(orc_binary)(out, arg_len, arg, arg_len, 1, 2);
    SFX(orc_fft)(out, points, 1);
    SFX(orc_real_scale)(out, arg_len, (REAL)1 / (REAL)points);
    SFX(orc_swap_halves)(out, arg_len, 1, 1);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Real-only interpolation between samples (time_freq/real_interpolation.rs:33-176).
 * dest_len = round((len-1)*factor) + 1; no wrap-around.  Index reads the reference would panic on
 * (delay pushing `before+1` past the end) are clamped to the last sample here.
 * ---------------------------------------------------------------------------------------- */
size_t SFX(orc_interpolate_real_len)(size_t len, REAL factor)
{
    return (size_t)R_ROUND((REAL)(len - 1) * factor) + 1;
}

static inline REAL SFX(at)(const REAL *x, size_t len, long long i)
{
    if (i < 0) i = 0;
    if ((size_t)i >= len) i = (long long)len - 1;
    return x[i];
}

void SFX(orc_interpolate_lin)(const REAL *x, size_t len, REAL factor, REAL delay, REAL *out)
{
    size_t dest_len = SFX(orc_interpolate_real_len)(len, factor);
    REAL i = 0;
    for (size_t n = 0; n + 1 < dest_len; ++n) {
        REAL rounded = i / factor + delay;
        REAL beforef = R_FLOOR(rounded);
        long long before = (long long)beforef;
        REAL y0 = SFX(at)(x, len, before), y1 = SFX(at)(x, len, before + 1);
        out[n] = y0 + (y1 - y0) * (rounded - beforef);
        i = i + (REAL)1;
    }
    out[dest_len - 1] = x[len - 1];
}

static REAL SFX(hermite)(REAL y0, REAL y1, REAL y2, REAL y3, REAL x)
{
    const REAL half = (REAL)0.5, c15 = (REAL)1.5, two = (REAL)2, c25 = (REAL)2.5;
    REAL x2 = x * x;
    REAL a0 = -half * y0 + c15 * y1 - c15 * y2 + half * y3;
    REAL a1 = y0 - c25 * y1 + two * y2 - half * y3;
    REAL a2 = -half * y0 + half * y2;
    REAL a3 = y1;
    return (a0 * x * x2) + (a1 * x2) + (a2 * x) + a3;
}

void SFX(orc_interpolate_hermite)(const REAL *x, size_t len, REAL factor, REAL delay, REAL *out)
{
    size_t dest_len = SFX(orc_interpolate_real_len)(len, factor);
    size_t start = (size_t)(-R_FLOOR(-(((REAL)1 - delay) * factor))); /* ceil */
    size_t end = start + 1;
    if (start > dest_len) start = dest_len;
    size_t tail = dest_len > end ? dest_len - end : 0;
    if (tail < start) tail = start;
    REAL i = 0;
    for (size_t n = 0; n < dest_len; ++n) {
        REAL rounded = i / factor + delay;
        REAL beforef = R_FLOOR(rounded);
        long long before = (long long)beforef;
        REAL xf = rounded - beforef;
        REAL y0, y1, y2, y3;
        if (n < start) { /* :103-124 first interval: y0 extrapolated */
            y1 = SFX(at)(x, len, before); y2 = SFX(at)(x, len, before + 1); y3 = SFX(at)(x, len, before + 2);
            y0 = y1 - (y2 - y1);
        } else if (n < tail) { /* :126-145 */
            y0 = SFX(at)(x, len, before - 1); y1 = SFX(at)(x, len, before);
            y2 = SFX(at)(x, len, before + 1); y3 = SFX(at)(x, len, before + 2);
        } else { /* :147-172 last intervals: y2/y3 extrapolated when past the end */
            y0 = SFX(at)(x, len, before - 1); y1 = SFX(at)(x, len, before);
            y2 = (before >= 0 && (size_t)before < len - 1) ? x[before + 1] : y1 + (y1 - y0);
            y3 = (before >= 0 && (size_t)before + 2 < len) ? x[before + 2] : y2 + (y2 - y1);
        }
        out[n] = SFX(hermite)(y0, y1, y2, y3, xf);
        i = i + (REAL)1;
    }
}

/* ------------------------------------------------------------------------------------------
 * Statistics, sums and dot products (vector/src/vector_types/general/statistics.rs:181-530,
 * dot_products.rs:67-165), single chunk (the default MultiCoreSettings).
 * out[0..7] real: sum, count, average, rms, min, min_index, max, max_index
 * complex: sum.re, sum.im, count, avg.re, avg.im, rms.re, rms.im, min.re, min.im, min_index,
 *          max.re, max.im, max_index                                    (13 doubles)
 * Element j of the walk is x[first + j*step]; indices reported are j (statistics_split uses
 * first = bucket, step = len, :399-426).
 * ---------------------------------------------------------------------------------------- */
void SFX(orc_real_statistics)(const REAL *x, size_t len, size_t first, size_t step, double *out)
{
    REAL sum = 0, sq = 0, mn = (REAL)INFINITY, mx = -(REAL)INFINITY;
    size_t cnt = 0, imn = 0, imx = 0;
    for (size_t i = first, j = 0; i < len; i += step, ++j) {
        REAL e = x[i];
        sum = sum + e; cnt += 1; sq = sq + e * e;
        if (e > mx) { mx = e; imx = j; }
        if (e < mn) { mn = e; imn = j; }
    }
    out[0] = sum; out[1] = (double)cnt; out[2] = sum / (REAL)cnt; out[3] = R_SQRT(sq / (REAL)cnt);
    out[4] = mn; out[5] = (double)imn; out[6] = mx; out[7] = (double)imx;
}

static void SFX(csqrt_polar)(REAL re, REAL im, REAL *ore, REAL *oim)
{
    /* num-complex 0.4 Complex::sqrt for finite inputs: principal root; purely real / imaginary inputs are special
     * cased there, the general branch is from_polar(sqrt(r), theta / 2) */
    if (im == 0) {
        if (re >= 0) { *ore = R_SQRT(re); *oim = im; } else { *ore = 0; *oim = im < 0 || (1 / im) < 0 ? -R_SQRT(-re) : R_SQRT(-re); }
        return;
    }
    if (re == 0) {
        REAL x = R_SQRT(R_FABS(im) / (REAL)2);
        *ore = x; *oim = im > 0 ? x : -x;
        return;
    }
    REAL r = R_HYPOT(re, im), th = R_ATAN2(im, re);
    *ore = R_SQRT(r) * R_COS(th / (REAL)2);
    *oim = R_SQRT(r) * R_SIN(th / (REAL)2);
}

void SFX(orc_complex_statistics)(const REAL *x, size_t len, size_t first, size_t step, double *out)
{
    size_t points = len / 2;
    REAL sr = 0, si = 0, qr = 0, qi = 0;
    REAL mnr = (REAL)INFINITY, mni = (REAL)INFINITY, mxr = 0, mxi = 0;
    size_t cnt = 0, imn = 0, imx = 0;
    for (size_t i = first, j = 0; i < points; i += step, ++j) {
        REAL re = x[2 * i], im = x[2 * i + 1];
        sr = sr + re; si = si + im; cnt += 1;
        qr = qr + (re * re - im * im); qi = qi + (re * im + im * re);
        if (R_HYPOT(re, im) > R_HYPOT(mxr, mxi)) { mxr = re; mxi = im; imx = j; }
        if (R_HYPOT(re, im) < R_HYPOT(mnr, mni)) { mnr = re; mni = im; imn = j; }
    }
    REAL rr, ri;
    SFX(csqrt_polar)(qr / (REAL)cnt, qi / (REAL)cnt, &rr, &ri);
    out[0] = sr; out[1] = si; out[2] = (double)cnt; out[3] = sr / (REAL)cnt; out[4] = si / (REAL)cnt;
    out[5] = rr; out[6] = ri; out[7] = mnr; out[8] = mni; out[9] = (double)imn; out[10] = mxr; out[11] = mxi;
    out[12] = (double)imx;
}

/* sum / sum_sq: out[0..1] (real: out[0]); dot: over min(len) scalars (real) or pairs (complex, no conjugation) */
void SFX(orc_sum)(const REAL *x, size_t len, int is_complex, int squared, double *out)
{
    REAL a = 0, b = 0;
    if (!is_complex) {
        for (size_t i = 0; i < len; ++i) a = a + (squared ? x[i] * x[i] : x[i]);
    } else {
        for (size_t i = 0; i + 1 < len; i += 2) {
            REAL re = x[i], im = x[i + 1];
            if (squared) { a = a + (re * re - im * im); b = b + (re * im + im * re); }
            else { a = a + re; b = b + im; }
        }
    }
    out[0] = a; out[1] = b;
}

void SFX(orc_dot)(const REAL *x, const REAL *y, size_t len, int is_complex, double *out)
{
    REAL a = 0, b = 0;
    if (!is_complex) {
        for (size_t i = 0; i < len; ++i) a = a + x[i] * y[i];
    } else {
        for (size_t i = 0; i + 1 < len; i += 2) {
            a = a + (x[i] * y[i] - x[i + 1] * y[i + 1]);
            b = b + (x[i] * y[i + 1] + x[i + 1] * y[i]);
        }
    }
    out[0] = a; out[1] = b;
}
