// elementwise.hip -- coalesced 16-byte-per-lane elementwise kernels (real and complex arithmetic,
// complex->real maps, windows).  Built with -ffp-contract=off: the reference performs each
// multiply and add as a separately rounded IEEE operation (Rust never contracts to FMA), so these
// kernels reproduce vector/src/vector_types/general/elementary.rs:283-360, 540-589 and
// complex/complex_ops.rs:81-116 bit for bit wherever the math library is not involved.
#include "bdsp_internal.h"
#include "dsp_funcs.h"
#include "ew_map.h"

namespace bdsp {

// ---- ops -------------------------------------------------------------------------------------
template <typename T> struct OpRealScale {
    struct Params { T f; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t, Params p)
    {
#pragma unroll 4
        for (int i = 0; i < n; ++i) e[i] = e[i] * p.f; // elementary.rs:327-342
    }
};
template <typename T> struct OpRealOffset {
    struct Params { T f; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t, Params p)
    {
#pragma unroll 4
        for (int i = 0; i < n; ++i) e[i] = e[i] + p.f; // elementary.rs:298-304
    }
};
template <typename T> struct OpComplexOffset {
    struct Params { T re, im; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t, Params p)
    {
#pragma unroll 2
        for (int i = 0; i + 1 < n; i += 2) { e[i] = e[i] + p.re; e[i + 1] = e[i + 1] + p.im; }
    }
};
template <typename T> struct OpComplexScale {
    struct Params { T re, im; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t, Params p)
    {
#pragma unroll 2
        for (int i = 0; i + 1 < n; i += 2) {
            T a = e[i], b = e[i + 1];
            e[i] = a * p.re - b * p.im; // num-complex Mul, elementary.rs:344-360
            e[i + 1] = a * p.im + b * p.re;
        }
    }
};
template <typename T> struct OpConj {
    struct Params { int unused; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t, Params)
    {
#pragma unroll 2
        for (int i = 1; i < n; i += 2) e[i] = -e[i]; // complex_ops.rs:107-116
    }
};
// multiply_complex_exponential: z[k] *= exp(j*(a*k + b)) with a, b already multiplied by delta
// (complex_ops.rs:81-105).  The reference advances a running product per element, whose error
// grows with the index; here every element gets its own phase, reduced in double, so the result
// is at least as close to the exact value as the reference's (compared with tolerance).
template <typename T> struct OpMulCexp {
    struct Params { double a, b; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t i0, Params p)
    {
        for (int i = 0; i + 1 < n; i += 2) {
            double k = (double)((i0 + i) / 2);
            double s, c;
            sincos(p.a * k + p.b, &s, &c);
            T wr = (T)c, wi = (T)s, zr = e[i], zi = e[i + 1];
            e[i] = zr * wr - zi * wi;
            e[i + 1] = zr * wi + zi * wr;
        }
    }
};
template <typename T> struct OpWindow {
    struct Params { int id; T alpha; size_t points; int is_complex; int unapply; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t i0, Params p)
    {
        // time.rs:32-66 -> vector_types/mod.rs:526-597
        if (p.is_complex) {
            for (int i = 0; i + 1 < n; i += 2) {
                T w = window_value_sym<T>(p.id, p.alpha, (i0 + i) / 2, p.points);
                if (p.unapply) w = (T)1 / w;
                T re = e[i], im = e[i + 1];
                e[i] = re * w - im * (T)0; // Complex * Complex::new(w, 0)
                e[i + 1] = re * (T)0 + im * w;
            }
        } else {
            for (int i = 0; i < n; ++i) {
                T w = window_value_sym<T>(p.id, p.alpha, i0 + i, p.points);
                if (p.unapply) w = (T)1 / w;
                e[i] = e[i] * w;
            }
        }
    }
};
// multiply_function_priv, symmetric functions (time_freq/mod.rs:655-721): element i takes the value
// of the function on the negative half of the axis, j = -|i - center|.
template <typename T> struct OpFreqResp {
    struct Params { int id; T rolloff; T ratio; size_t points; int is_complex; int shifted; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t i0, Params p)
    {
        const size_t offset = p.points % 2;
        const T maxv = (T)(p.points - offset) / (T)2;
        const int step = p.is_complex ? 2 : 1;
        for (int i = 0; i + step - 1 < n; i += step) {
            T j = -maxv + (T)((i0 + i) / step);
            if (j > (T)0) j = -j;
            T arg = p.ratio * conv_freq_value<T>(p.id, p.rolloff, fft_swap_x<T>(p.shifted != 0, j, maxv) * p.ratio);
            if (p.is_complex) {
                T re = e[i], im = e[i + 1];
                e[i] = re * arg - im * (T)0;
                e[i + 1] = re * (T)0 + im * arg;
            } else e[i] = e[i] * arg;
        }
    }
};
// apply_linear_phase (interpolation.rs:319-339): bins below pos_points get phase_inc*k, the rest
// phase_inc*(k - points); every element evaluates its own phase (the reference runs two running
// products, compared with tolerance).
template <typename T> struct OpLinearPhase {
    struct Params { double phase_inc; size_t points; size_t pos_points; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t i0, Params p)
    {
        for (int i = 0; i + 1 < n; i += 2) {
            size_t k = (i0 + i) / 2;
            double kk = k < p.pos_points ? (double)k : (double)k - (double)p.points;
            double s, c;
            sincos(p.phase_inc * kk, &s, &c);
            T wr = (T)c, wi = (T)s, zr = e[i], zi = e[i + 1];
            e[i] = zr * wr - zi * wi;
            e[i + 1] = zr * wi + zi * wr;
        }
    }
};
template <typename T> struct OpFill {
    struct Params { T v; };
    static __device__ __forceinline__ void apply(T* e, int n, size_t, Params p)
    {
        for (int i = 0; i < n; ++i) e[i] = p.v;
    }
};

template <typename T> int ew_real_scale(T* x, size_t len, T f, hipStream_t s)
{ return launch_map<T, OpRealScale<T>>(x, len, {f}, s); }
template <typename T> int ew_real_offset(T* x, size_t len, bool is_complex, T f, hipStream_t s)
{
    // a real offset on a complex vector adds (f, 0) to every point (elementary.rs:291-297)
    if (is_complex) return launch_map<T, OpComplexOffset<T>>(x, len, {f, (T)0}, s);
    return launch_map<T, OpRealOffset<T>>(x, len, {f}, s);
}
template <typename T> int ew_complex_scale(T* x, size_t len, T re, T im, hipStream_t s)
{ return launch_map<T, OpComplexScale<T>>(x, len, {re, im}, s); }
template <typename T> int ew_complex_offset(T* x, size_t len, T re, T im, hipStream_t s)
{ return launch_map<T, OpComplexOffset<T>>(x, len, {re, im}, s); }
template <typename T> int ew_conj(T* x, size_t len, hipStream_t s)
{ return launch_map<T, OpConj<T>>(x, len, {0}, s); }
template <typename T> int ew_mul_cexp(T* x, size_t len, T a, T b, hipStream_t s)
{ return launch_map<T, OpMulCexp<T>>(x, len, {(double)a, (double)b}, s); }
template <typename T> int ew_window(T* x, size_t len, bool is_complex, int id, T alpha, bool unapply, hipStream_t s)
{
    size_t points = is_complex ? len / 2 : len;
    return launch_map<T, OpWindow<T>>(x, len, {id, alpha, points, (int)is_complex, (int)unapply}, s);
}
template <typename T> int ew_fill(T* x, size_t len, T value, hipStream_t s)
{ return launch_map<T, OpFill<T>>(x, len, {value}, s); }
template <typename T> int ew_freq_response(T* x, size_t len, bool is_complex, int fid, T rolloff, T ratio, bool shifted, hipStream_t s)
{
    size_t points = is_complex ? len / 2 : len;
    return launch_map<T, OpFreqResp<T>>(x, len, {fid, rolloff, ratio, points, (int)is_complex, (int)shifted}, s);
}
template <typename T> int ew_linear_phase(T* x, size_t len, T delay, hipStream_t s)
{
    // phase_inc = 2*pi*delay/points computed in T like the reference, then widened
    size_t points = len / 2, pos = points / 2;
    T phase_inc = (T)2 * (T)3.14159265358979323846 * delay / (T)points;
    return launch_map<T, OpLinearPhase<T>>(x, len, {(double)phase_inc, points, pos}, s);
}

// ---- spectrum resampling for the FFT-domain interpolation family (round 4) ------------------------
// out[k], k < dst_points, from an N-point spectrum `in` (complex, out != in), in ONE trip instead of three or four:
//   MODE 0 (interpolatei): out[k] = in[k mod N] -- the transform of a vector interleaved with factor - 1 zeros IS the
//          factor-fold periodic repetition of the transform of the vector itself (sum_i x[i] e^{-2 pi i (i f) k / (f N)} =
//          X[k mod N]), so the reference's zero_interleave -> plain_fft of f N points (interpolation.rs:484-532) is an
//          N-point transform read f times;
//   MODE 1 (interpolate / interpft, upsampling): zero_pad(Center) -- the first ceil(N/2) bins stay, the last floor(N/2)
//          move to the end, zeros between (data_reorganization.rs:343-358) -- with apply_linear_phase (:319-339) on the
//          SOURCE bin when delay != 0;
//   MODE 2 (interpolate, downsampling): the crop of interpolate_downsample with the same linear phase;
// then x the frequency response on the destination axis (OpFreqResp: the same formula, so the multiplier is bit-identical
// to the separate pass), or x `ratio` alone (fid < 0), or nothing (fid == -2: a host-sampled response follows).
// Every product is rounded to T in the same order as the separate passes (this file is compiled without contraction).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_spectrum_resample(const T* __restrict__ in, T* __restrict__ out, size_t src_points, size_t dst_points,
                                                            int fid, T rolloff, T ratio, double phase_inc)
{
    const size_t offset = dst_points % 2;
    const T maxv = (T)(dst_points - offset) / (T)2;
    const size_t pos = src_points - src_points / 2, neg = src_points / 2; // bins that stay / move to the end
    const size_t ph_pos = src_points / 2;                                  // OpLinearPhase's positive bins
    typedef T vec2 __attribute__((ext_vector_type(2)));
    const vec2* in2 = reinterpret_cast<const vec2*>(in);
    vec2* out2 = reinterpret_cast<vec2*>(out);
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < dst_points; k += (size_t)gridDim.x * blockDim.x) {
        size_t sk;
        bool zero = false;
        if (MODE == 0) sk = k % src_points;
        else if (MODE == 2) { // interpolate_downsample (interpolation.rs:362-376): the first ceil(D/2) and the last floor(D/2) bins
            sk = k < dst_points - dst_points / 2 ? k : k + (src_points - dst_points);
        } else {
            if (k < pos) sk = k;
            else if (k >= dst_points - neg) sk = k - (dst_points - src_points);
            else { sk = 0; zero = true; }
        }
        T re = (T)0, im = (T)0;
        if (!zero) {
            const vec2 z = in2[sk];
            re = z.x;
            im = z.y;
            if (MODE != 0 && phase_inc != 0.0) { // OpLinearPhase on the source bin
                const double kk = sk < ph_pos ? (double)sk : (double)sk - (double)src_points;
                double sn, cs;
                sincos(phase_inc * kk, &sn, &cs);
                const T wr = (T)cs, wi = (T)sn, zr = re, zi = im;
                re = zr * wr - zi * wi;
                im = zr * wi + zi * wr;
            }
        }
        if (fid >= 0) { // OpFreqResp on the destination axis
            T j = -maxv + (T)k;
            if (j > (T)0) j = -j;
            const T arg = ratio * conv_freq_value<T>(fid, rolloff, fft_swap_x<T>(true, j, maxv) * ratio);
            const T r2 = re * arg - im * (T)0, i2 = re * (T)0 + im * arg;
            re = r2; im = i2;
        } else if (fid == -1) {
            re = re * ratio; im = im * ratio;
        }
        out2[k] = vec2{re, im};
    }
}
template <typename T>
int ew_spectrum_resample(const T* in, T* out, size_t src_points, size_t dst_points, int mode, int fid, T rolloff, T ratio, double phase_inc, hipStream_t s)
{
    if (dst_points == 0) return BDSP_OK;
    if (in == out || src_points == 0) return BDSP_ERR_UNSUPPORTED;
    const unsigned grid = ew_grid(dst_points / 2 + 1);
    if (mode == 0) hipLaunchKernelGGL((k_spectrum_resample<T, 0>), dim3(grid), dim3(256), 0, s, in, out, src_points, dst_points, fid, rolloff, ratio, phase_inc);
    else if (mode == 2) hipLaunchKernelGGL((k_spectrum_resample<T, 2>), dim3(grid), dim3(256), 0, s, in, out, src_points, dst_points, fid, rolloff, ratio, phase_inc);
    else hipLaunchKernelGGL((k_spectrum_resample<T, 1>), dim3(grid), dim3(256), 0, s, in, out, src_points, dst_points, fid, rolloff, ratio, phase_inc);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- binary vector (.) vector, in place on x (elementary.rs:540-589) ------------------------------
template <typename T, int OP, bool CPLX>
__global__ __launch_bounds__(256) void k_binary(T* __restrict__ x, const T* __restrict__ y, size_t len)
{
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    const size_t nvec = len / VN;
    V* xv = reinterpret_cast<V*>(x);
    const V* yv = reinterpret_cast<const V*>(y);
    auto op = [](T* a, const T* b, int n) {
        if (!CPLX || OP < 2) {
            for (int i = 0; i < n; ++i) {
                if (OP == 0) a[i] = a[i] + b[i];
                else if (OP == 1) a[i] = a[i] - b[i];
                else if (OP == 2) a[i] = a[i] * b[i];
                else a[i] = a[i] / b[i];
            }
        } else {
            for (int i = 0; i + 1 < n; i += 2) {
                T ar = a[i], ai = a[i + 1], br = b[i], bi = b[i + 1];
                if (OP == 2) {
                    a[i] = ar * br - ai * bi;
                    a[i + 1] = ar * bi + ai * br;
                } else { // num-complex Div
                    T nn = br * br + bi * bi;
                    a[i] = (ar * br + ai * bi) / nn;
                    a[i + 1] = (ai * br - ar * bi) / nn;
                }
            }
        }
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        V pa = xv[i];
        V pb = yv[i];
        op(reinterpret_cast<T*>(&pa), reinterpret_cast<const T*>(&pb), VN);
        xv[i] = pa;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        size_t done = nvec * VN;
        if (done < len) op(x + done, y + done, (int)(len - done));
    }
}

template <typename T> int ew_binary(T* x, const T* y, size_t len, bool is_complex, int op, hipStream_t s)
{
    if (len == 0) return BDSP_OK;
    dim3 grid(ew_grid(len / Vec16<T>::N + 1)), block(256);
#define BDSP_BIN(OPV)                                                                              \
    do {                                                                                           \
        if (is_complex) hipLaunchKernelGGL((k_binary<T, OPV, true>), grid, block, 0, s, x, y, len); \
        else hipLaunchKernelGGL((k_binary<T, OPV, false>), grid, block, 0, s, x, y, len);           \
    } while (0)
    switch (op) {
    case 0: BDSP_BIN(0); break;
    case 1: BDSP_BIN(1); break;
    case 2: BDSP_BIN(2); break;
    default: BDSP_BIN(3); break;
    }
#undef BDSP_BIN
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- x[i] (.)= y[i mod ylen]: add_smaller / sub_smaller / mul_smaller / div_smaller (elementary.rs:591-640)
template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_binary_smaller(T* __restrict__ x, const T* __restrict__ y, size_t points,
                                                         size_t ypoints, int op)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < points; i += (size_t)gridDim.x * blockDim.x) {
        size_t j = i % ypoints;
        if (CPLX) {
            T ar = x[2 * i], ai = x[2 * i + 1], br = y[2 * j], bi = y[2 * j + 1];
            T re, im;
            if (op == 0) { re = ar + br; im = ai + bi; }
            else if (op == 1) { re = ar - br; im = ai - bi; }
            else if (op == 2) { re = ar * br - ai * bi; im = ar * bi + ai * br; }
            else { T nn = br * br + bi * bi; re = (ar * br + ai * bi) / nn; im = (ai * br - ar * bi) / nn; }
            x[2 * i] = re; x[2 * i + 1] = im;
        } else {
            T a = x[i], b = y[j];
            x[i] = op == 0 ? a + b : (op == 1 ? a - b : (op == 2 ? a * b : a / b));
        }
    }
}
template <typename T> int ew_binary_smaller(T* x, const T* y, size_t len, size_t ylen, bool is_complex, int op, hipStream_t s)
{
    const size_t e = is_complex ? 2 : 1, points = len / e, yp = ylen / e;
    if (points == 0 || yp == 0) return BDSP_OK;
    if (is_complex) hipLaunchKernelGGL((k_binary_smaller<T, true>), dim3(ew_grid(points)), dim3(256), 0, s, x, y, points, yp, op);
    else hipLaunchKernelGGL((k_binary_smaller<T, false>), dim3(ew_grid(points)), dim3(256), 0, s, x, y, points, yp, op);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- x[point i] *= (or /=) table[i]: host-sampled callback windows / frequency responses ------------
// (interop/src/lib.rs:245-377: the C callbacks cannot run on the device, so the host samples them once)
template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_point_table(T* __restrict__ x, const T* __restrict__ table, size_t points, bool divide)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < points; i += (size_t)gridDim.x * blockDim.x) {
        T w = table[i];
        if (CPLX) {
            T re = x[2 * i], im = x[2 * i + 1];
            x[2 * i] = divide ? re / w : re * w;
            x[2 * i + 1] = divide ? im / w : im * w;
        } else {
            x[i] = divide ? x[i] / w : x[i] * w;
        }
    }
}
template <typename T> int ew_point_table(T* x, size_t len, bool is_complex, const T* table, bool divide, hipStream_t s)
{
    const size_t points = is_complex ? len / 2 : len;
    if (points == 0) return BDSP_OK;
    if (is_complex) hipLaunchKernelGGL((k_point_table<T, true>), dim3(ew_grid(points)), dim3(256), 0, s, x, table, points, divide);
    else hipLaunchKernelGGL((k_point_table<T, false>), dim3(ew_grid(points)), dim3(256), 0, s, x, table, points, divide);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- complex -> real (complex_to_real.rs:374-478); out may alias x (in-place compaction) -----------
template <typename T> __device__ __forceinline__ T dev_hypot2(T a, T b);
template <> __device__ __forceinline__ float dev_hypot2<float>(float a, float b) { return hypotf(a, b); }
template <> __device__ __forceinline__ double dev_hypot2<double>(double a, double b) { return hypot(a, b); }
template <typename T> __device__ __forceinline__ T dev_atan2(T a, T b);
template <> __device__ __forceinline__ float dev_atan2<float>(float a, float b) { return atan2f(a, b); }
template <> __device__ __forceinline__ double dev_atan2<double>(double a, double b) { return atan2(a, b); }

template <typename T>
__device__ __forceinline__ T c2r_one(cpx<T> z, int kind)
{
    switch (kind) {
    case 0: return dev_hypot2<T>(z.x, z.y);
    case 1: return z.x * z.x + z.y * z.y;
    case 2: return z.x;
    case 3: return z.y;
    default: return dev_atan2<T>(z.y, z.x);
    }
}

// two points per lane: one 2-element load, one 2-scalar store
template <typename T>
__global__ __launch_bounds__(256) void k_complex_to_real(const cpx<T>* __restrict__ x, T* __restrict__ out,
                                                          size_t points, int kind)
{
    struct alignas(2 * sizeof(cpx<T>)) C2 { cpx<T> a, b; };
    struct alignas(2 * sizeof(T)) R2 { T a, b; };
    const size_t pairs = points / 2;
    const C2* x2 = reinterpret_cast<const C2*>(x);
    R2* o2 = reinterpret_cast<R2*>(out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs;
         i += (size_t)gridDim.x * blockDim.x) {
        C2 z = x2[i];
        o2[i] = R2{c2r_one<T>(z.a, kind), c2r_one<T>(z.b, kind)};
    }
    if ((points & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[points - 1] = c2r_one<T>(x[points - 1], kind);
}

template <typename T> int ew_complex_to_real(const T* x, T* out, size_t len, int kind, hipStream_t s)
{
    size_t points = len / 2;
    if (points == 0) return BDSP_OK;
    if (static_cast<const void*>(x) == static_cast<const void*>(out)) {
        set_last_error("complex_to_real: in-place compaction is not race free on a GPU; pass the trade buffer");
        return BDSP_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((k_complex_to_real<T>), dim3(ew_grid(points)), dim3(256), 0, s,
                       reinterpret_cast<const cpx<T>*>(x), out, points, kind);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

#define BDSP_INST(T)                                                                               \
    template int ew_real_scale<T>(T*, size_t, T, hipStream_t);                                     \
    template int ew_real_offset<T>(T*, size_t, bool, T, hipStream_t);                              \
    template int ew_complex_scale<T>(T*, size_t, T, T, hipStream_t);                               \
    template int ew_complex_offset<T>(T*, size_t, T, T, hipStream_t);                              \
    template int ew_binary<T>(T*, const T*, size_t, bool, int, hipStream_t);                       \
    template int ew_binary_smaller<T>(T*, const T*, size_t, size_t, bool, int, hipStream_t);          \
    template int ew_point_table<T>(T*, size_t, bool, const T*, bool, hipStream_t);                   \
    template int ew_conj<T>(T*, size_t, hipStream_t);                                              \
    template int ew_mul_cexp<T>(T*, size_t, T, T, hipStream_t);                                    \
    template int ew_complex_to_real<T>(const T*, T*, size_t, int, hipStream_t);                    \
    template int ew_window<T>(T*, size_t, bool, int, T, bool, hipStream_t);                        \
    template int ew_fill<T>(T*, size_t, T, hipStream_t);                                            \
    template int ew_freq_response<T>(T*, size_t, bool, int, T, T, bool, hipStream_t);              \
    template int ew_linear_phase<T>(T*, size_t, T, hipStream_t);                                    \
    template int ew_spectrum_resample<T>(const T*, T*, size_t, size_t, int, int, T, T, double, hipStream_t);
BDSP_INST(float)
BDSP_INST(double)
#undef BDSP_INST

} // namespace bdsp
