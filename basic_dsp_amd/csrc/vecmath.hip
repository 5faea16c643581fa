// vecmath.hip -- the facade's remaining per-element families, device resident:
//   * TrigOps / PowerOps / abs / wrap / the "approximated" ops (trigonometry_and_powers.rs:196-420,
//     real_ops.rs:236-375) as one in-place packet map,
//   * diff / diff_with_start / cum_sum (diff_sum.rs:65-122), unwrap (real_ops.rs:262-284),
//   * get/set real_imag and mag_phase (complex_to_real.rs:674-770), split_into / merge
//     (data_reorganization.rs:484-555).
// The complex functions restate num-complex 0.4's formulas (polar forms of sqrt/powf/ln/log/expf, logarithmic
// forms of the inverse functions) with every multiply and add rounded separately (-ffp-contract=off); what differs
// from the reference is only the last-ulp behaviour of the math library underneath.
#include "bdsp_internal.h"
#include "ew_map.h"

namespace bdsp {

template <typename T> struct Cx { T re, im; };
template <typename T> __device__ __forceinline__ Cx<T> cx(T re, T im) { return Cx<T>{re, im}; }
template <typename T> __device__ __forceinline__ Cx<T> cx_mul(Cx<T> a, Cx<T> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename T> __device__ __forceinline__ Cx<T> cx_add(Cx<T> a, Cx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> __device__ __forceinline__ Cx<T> cx_sub(Cx<T> a, Cx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <typename T> __device__ __forceinline__ Cx<T> cx_div(Cx<T> a, Cx<T> b)
{
    const T n = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / n, (a.im * b.re - a.re * b.im) / n};
}
template <typename T> __device__ __forceinline__ Cx<T> cx_from_polar(T r, T t)
{
    T s, c;
    sincos(t, &s, &c);
    return {r * c, r * s};
}
template <> __device__ __forceinline__ Cx<float> cx_from_polar<float>(float r, float t)
{
    float s, c;
    sincosf(t, &s, &c);
    return {r * c, r * s};
}
template <typename T> __device__ __forceinline__ Cx<T> cx_ln(Cx<T> z) { return {log(hypot(z.re, z.im)), atan2(z.im, z.re)}; }
template <typename T> __device__ __forceinline__ Cx<T> cx_sqrt(Cx<T> z)
{
    if (z.im == T(0)) {
        if (!signbit(z.re)) return {sqrt(z.re), z.im};
        const T im = sqrt(-z.re);
        return {T(0), signbit(z.im) ? -im : im};
    }
    if (z.re == T(0)) {
        const T x = sqrt(fabs(z.im) / T(2));
        return {x, signbit(z.im) ? -x : x};
    }
    return cx_from_polar<T>(sqrt(hypot(z.re, z.im)), atan2(z.im, z.re) / T(2));
}

template <typename T> __device__ __forceinline__ Cx<T> cx_apply(Cx<T> z, int fn, T arg)
{
    const Cx<T> one{T(1), T(0)}, two{T(2), T(0)}, i{T(0), T(1)}, mi{T(0), T(-1)};
    switch (fn) {
    case MATH_SQRT: return cx_sqrt(z);
    case MATH_SQUARE: return cx_mul(z, z);
    case MATH_POWF:
        if (arg == T(0)) return one;
        return cx_from_polar<T>(pow(hypot(z.re, z.im), arg), atan2(z.im, z.re) * arg);
    case MATH_LN: return cx_ln(z);
    case MATH_EXP: return cx_from_polar<T>(exp(z.re), z.im);
    case MATH_LOG: return {log(hypot(z.re, z.im)) / log(arg), atan2(z.im, z.re) / log(arg)};
    case MATH_EXPF: return cx_from_polar<T>(pow(arg, z.re), z.im * log(arg));
    case MATH_SIN: return {sin(z.re) * cosh(z.im), cos(z.re) * sinh(z.im)};
    case MATH_COS: return {cos(z.re) * cosh(z.im), -sin(z.re) * sinh(z.im)};
    case MATH_TAN: {
        const T a = z.re + z.re, b = z.im + z.im, d = cos(a) + cosh(b);
        return {sin(a) / d, sinh(b) / d};
    }
    case MATH_ASIN: return cx_mul(mi, cx_ln(cx_add(cx_sqrt(cx_sub(one, cx_mul(z, z))), cx_mul(i, z))));
    case MATH_ACOS: return cx_mul(mi, cx_ln(cx_add(cx_mul(i, cx_sqrt(cx_sub(one, cx_mul(z, z)))), z)));
    case MATH_ATAN:
        if (z.re == T(0) && z.im == T(1)) return {T(0), (T)INFINITY};
        if (z.re == T(0) && z.im == T(-1)) return {T(0), -(T)INFINITY};
        return cx_div(cx_sub(cx_ln(cx_add(one, cx_mul(i, z))), cx_ln(cx_sub(one, cx_mul(i, z)))), cx_mul(two, i));
    case MATH_SINH: return {sinh(z.re) * cos(z.im), cosh(z.re) * sin(z.im)};
    case MATH_COSH: return {cosh(z.re) * cos(z.im), sinh(z.re) * sin(z.im)};
    case MATH_TANH: {
        const T a = z.re + z.re, b = z.im + z.im, d = cosh(a) + cos(b);
        return {sinh(a) / d, sin(b) / d};
    }
    case MATH_ASINH: return cx_ln(cx_add(z, cx_sqrt(cx_add(one, cx_mul(z, z)))));
    case MATH_ACOSH:
        return cx_mul(two, cx_ln(cx_add(cx_sqrt(cx_div(cx_add(z, one), two)), cx_sqrt(cx_div(cx_sub(z, one), two)))));
    case MATH_ATANH:
        if (z.re == T(1) && z.im == T(0)) return {(T)INFINITY, T(0)};
        if (z.re == T(-1) && z.im == T(0)) return {-(T)INFINITY, T(0)};
        return cx_div(cx_sub(cx_ln(cx_add(one, z)), cx_ln(cx_sub(one, z))), two);
    default: return z;
    }
}

template <typename T> __device__ __forceinline__ T re_apply(T x, int fn, T arg)
{
    switch (fn) {
    case MATH_SQRT: return sqrt(x);
    case MATH_SQUARE: return x * x;
    case MATH_POWF: return pow(x, arg);
    case MATH_LN: return log(x);
    case MATH_EXP: return exp(x);
    case MATH_LOG: return log(x) / log(arg);
    case MATH_EXPF: return pow(arg, x);
    case MATH_SIN: return sin(x);
    case MATH_COS: return cos(x);
    case MATH_TAN: return tan(x);
    case MATH_ASIN: return asin(x);
    case MATH_ACOS: return acos(x);
    case MATH_ATAN: return atan(x);
    case MATH_SINH: return sinh(x);
    case MATH_COSH: return cosh(x);
    case MATH_TANH: return tanh(x);
    case MATH_ASINH: return asinh(x);
    case MATH_ACOSH: return acosh(x);
    case MATH_ATANH: return atanh(x);
    case MATH_ABS: return fabs(x);
    case MATH_WRAP: return fmod(x, arg);
    case MATH_EXPF_APPROX: return exp(log(arg) * x);
    case MATH_POWF_APPROX: return exp(log(x) * arg);
    default: return x;
    }
}

template <typename T> struct OpMath {
    struct Params { int fn; int cplx; T arg; };
    // One 16-byte packet held in REGISTERS: the 25-way switch is far too large to unroll per element, and a rolled loop
    // that indexes the packet with its counter would push the packet to scratch -- so the element is picked and put back
    // with compare-selects on the counter (VN <= 4).
    template <int VN>
    static __device__ __forceinline__ void apply_packet(T (&el)[VN], Params p)
    {
        const int step = p.cplx ? 2 : 1;
#pragma unroll 1
        for (int k = 0; k < VN; k += step) {
            T a = el[0], b = VN > 1 ? el[1] : T(0);
#pragma unroll
            for (int j = 1; j < VN; ++j) {
                a = (k == j) ? el[j] : a;
                if (j + 1 < VN) b = (k == j) ? el[j + 1] : b;
            }
            if (p.cplx) {
                const Cx<T> r = cx_apply<T>(Cx<T>{a, b}, p.fn, p.arg);
                a = r.re; b = r.im;
            } else {
                a = re_apply<T>(a, p.fn, p.arg);
            }
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                el[j] = (k == j) ? a : el[j];
                if (p.cplx && j > 0) el[j] = (k == j - 1) ? b : el[j];
            }
        }
    }
    static __device__ __forceinline__ void apply(T* e, int n, size_t, Params p)
    {
        if (p.cplx) {
            for (int i = 0; i + 1 < n; i += 2) {
                const Cx<T> r = cx_apply<T>(Cx<T>{e[i], e[i + 1]}, p.fn, p.arg);
                e[i] = r.re; e[i + 1] = r.im;
            }
        } else {
            for (int i = 0; i < n; ++i) e[i] = re_apply<T>(e[i], p.fn, p.arg);
        }
    }
};

// The math family is bound by its transcendental functions, not by memory: ONE packet per loop iteration (k_map_simple) with
// the 25-way switch inlined once, instead of k_map_inplace's four packets in flight -- as out-of-line calls the switch
// functions needed a stack frame (80 bytes of scratch per lane in the f64 instantiation, the last kernel of the library
// that used any).
template <typename T, typename OP>
__global__ __launch_bounds__(256) void k_map_simple(T* __restrict__ x, size_t len, typename OP::Params p)
{
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    const size_t nvec = len / VN;
    V* xv = reinterpret_cast<V*>(x);
    // contiguous runs per workgroup, like k_map_inplace (far-apart concurrent streams thrash DRAM pages)
    size_t per = (nvec + gridDim.x - 1) / gridDim.x;
    per = (per + 255) / 256 * 256;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = p0 + per < nvec ? p0 + per : nvec;
    for (size_t i = p0 + threadIdx.x; i < p1; i += 256) {
        V pk = xv[i];
        T el[VN];
#pragma unroll
        for (int j = 0; j < VN; ++j) el[j] = reinterpret_cast<T*>(&pk)[j];
        OP::template apply_packet<VN>(el, p);
#pragma unroll
        for (int j = 0; j < VN; ++j) reinterpret_cast<T*>(&pk)[j] = el[j];
        xv[i] = pk;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const size_t done = nvec * VN;
        if (done < len) OP::apply(x + done, (int)(len - done), done, p); // (fewer than VN scalars, in place)
    }
}

template <typename T> int ew_math(T* x, size_t len, bool is_complex, int fn, T arg, hipStream_t s)
{
    if (len == 0) return BDSP_OK;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0) {
        set_last_error("elementwise: buffer must be 16-byte aligned");
        return BDSP_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((k_map_simple<T, OpMath<T>>), dim3(ew_grid(len / Vec16<T>::N + 1)), dim3(256), 0, s, x, len,
                       typename OpMath<T>::Params{fn, is_complex ? 1 : 0, arg});
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- diff / diff_with_start: out-of-place into the trade buffer ------------------------------------------------
// out[j] = in[j + step] - in[j] (diff, `n` outputs) or out[j] = j < step ? in[j] : in[j] - in[j - step]
template <typename T>
__global__ __launch_bounds__(256) void k_diff(const T* __restrict__ in, T* __restrict__ out, size_t n, size_t step,
                                              bool with_start)
{
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (size_t)gridDim.x * 256) {
        if (with_start) out[j] = j < step ? in[j] : in[j] - in[j - step];
        else out[j] = in[j + step] - in[j];
    }
}
template <typename T> int vm_diff(const T* in, T* out, size_t n_out, size_t step, bool with_start, hipStream_t s)
{
    if (n_out == 0) return BDSP_OK;
    hipLaunchKernelGGL(k_diff<T>, dim3(ew_grid(n_out)), dim3(256), 0, s, in, out, n_out, step, with_start);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- cum_sum: chunk sums -> scan of the chunk sums -> rescan of every chunk with its offset ---------------------
// The running sum is carried in double whatever T is (the reference adds sequentially in T; the result here is the
// correctly rounded-once prefix, compared with tolerance).  E interleaved sequences (1 real, 2 complex).
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_CHUNK = 256 * SCAN_PER_THREAD; // elements per workgroup

// chunk sums: order does not matter, so the chunk is read with unit stride across the workgroup
template <typename T, int E>
__global__ __launch_bounds__(256) void k_scan_sums(const T* __restrict__ x, size_t n, double* __restrict__ sums)
{
    __shared__ double sh[E][256];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    double acc[E] = {};
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < n)
            for (int c = 0; c < E; ++c) acc[c] += (double)x[i * E + c];
    }
    for (int c = 0; c < E; ++c) sh[c][threadIdx.x] = acc[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int c = 0; c < E; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int c = 0; c < E; ++c) sums[(size_t)blockIdx.x * E + c] = sh[c][0];
}

// exclusive scan of the chunk sums by one workgroup (each thread owns a contiguous run)
template <int E>
__global__ __launch_bounds__(256) void k_scan_offsets(double* __restrict__ sums, size_t nchunks)
{
    __shared__ double sh[E][256];
    const size_t per = (nchunks + 255) / 256, b = threadIdx.x * per, e = b + per < nchunks ? b + per : nchunks;
    double acc[E] = {};
    for (size_t i = b; i < e; ++i)
        for (int c = 0; c < E; ++c) acc[c] += sums[i * E + c];
    for (int c = 0; c < E; ++c) sh[c][threadIdx.x] = acc[c];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int c = 0; c < E; ++c) {
            double run = 0.0;
            for (int t = 0; t < 256; ++t) { const double v = sh[c][t]; sh[c][t] = run; run += v; }
        }
    __syncthreads();
    double run[E];
    for (int c = 0; c < E; ++c) run[c] = sh[c][threadIdx.x];
    for (size_t i = b; i < e; ++i)
        for (int c = 0; c < E; ++c) { const double v = sums[i * E + c]; sums[i * E + c] = run[c]; run[c] += v; }
}

// A thread scans 16 CONSECUTIVE elements; the chunk travels global <-> LDS with unit stride across the workgroup
// (element e lives at LDS slot e + e/16, so the per-thread runs start in different banks).  Reading the runs
// straight from global memory -- every lane its own cache line -- measured 511 us for 16M complex f32 points.
template <typename T, int E>
__global__ __launch_bounds__(256) void k_scan_apply(T* __restrict__ x, size_t n, const double* __restrict__ offsets)
{
    struct El { T c[E]; };
    __shared__ El tile[SCAN_CHUNK + SCAN_CHUNK / 16];
    __shared__ double sh[E][256];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    El* xe = reinterpret_cast<El*>(x);
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        const int e = k * 256 + threadIdx.x;
        El v;
        for (int c = 0; c < E; ++c) v.c[c] = T(0);
        if (base + e < n) v = xe[base + e];
        tile[e + (e >> 4)] = v;
    }
    __syncthreads();
    double v[SCAN_PER_THREAD][E];
    double acc[E] = {};
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        const El el = tile[threadIdx.x * 17 + k];
        for (int c = 0; c < E; ++c) { v[k][c] = (double)el.c[c]; acc[c] += v[k][c]; }
    }
    for (int c = 0; c < E; ++c) sh[c][threadIdx.x] = acc[c];
    __syncthreads();
    // Hillis-Steele inclusive scan of the 256 thread totals
    for (int d = 1; d < 256; d <<= 1) {
        double t[E];
        for (int c = 0; c < E; ++c) t[c] = (int)threadIdx.x >= d ? sh[c][threadIdx.x - d] : 0.0;
        __syncthreads();
        for (int c = 0; c < E; ++c) sh[c][threadIdx.x] += t[c];
        __syncthreads();
    }
    double run[E];
    for (int c = 0; c < E; ++c) run[c] = offsets[(size_t)blockIdx.x * E + c] + (threadIdx.x ? sh[c][threadIdx.x - 1] : 0.0);
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        El el;
        for (int c = 0; c < E; ++c) { run[c] += v[k][c]; el.c[c] = (T)run[c]; }
        tile[threadIdx.x * 17 + k] = el;
    }
    __syncthreads();
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        const int e = k * 256 + threadIdx.x;
        if (base + e < n) xe[base + e] = tile[e + (e >> 4)];
    }
}

// `scratch` holds E doubles per chunk of SCAN_CHUNK elements
template <typename T> size_t vm_cum_sum_scratch(size_t len, bool is_complex)
{
    const size_t n = is_complex ? len / 2 : len;
    return sizeof(double) * 2 * ((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1);
}
template <typename T> int vm_cum_sum(T* x, size_t len, bool is_complex, void* scratch, hipStream_t s)
{
    const size_t n = is_complex ? len / 2 : len;
    if (n == 0) return BDSP_OK;
    const size_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    double* sums = static_cast<double*>(scratch);
    if (is_complex) {
        hipLaunchKernelGGL((k_scan_sums<T, 2>), dim3((unsigned)nchunks), dim3(256), 0, s, x, n, sums);
        hipLaunchKernelGGL((k_scan_offsets<2>), dim3(1), dim3(256), 0, s, sums, nchunks);
        hipLaunchKernelGGL((k_scan_apply<T, 2>), dim3((unsigned)nchunks), dim3(256), 0, s, x, n, sums);
    } else {
        hipLaunchKernelGGL((k_scan_sums<T, 1>), dim3((unsigned)nchunks), dim3(256), 0, s, x, n, sums);
        hipLaunchKernelGGL((k_scan_offsets<1>), dim3(1), dim3(256), 0, s, sums, nchunks);
        hipLaunchKernelGGL((k_scan_apply<T, 1>), dim3((unsigned)nchunks), dim3(256), 0, s, x, n, sums);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- unwrap (real_ops.rs:262-284) --------------------------------------------------------------------------------
// y[j] = F(x[j], y[j-1]) with a data-dependent branch on the ALREADY UNWRAPPED neighbour: a genuinely sequential
// recurrence (its state does not reduce to an associative operator), so one lane walks the vector while the
// wavefront stages tiles through LDS with coalesced packets.  Exact, not fast (see DESIGN.md for the rate).
// fmod on the serial critical path: the remainder a - trunc(a/b)*b is exactly representable, so one fma returns it
// exactly when the quotient is right; a quotient off by one (a/b rounded across an integer) shows as a remainder
// outside [0, |b|) and is redone.  Huge quotients, infinities and NaNs go to the library function.
template <typename T>
__device__ __forceinline__ T fmod_exact(T a, T b, T inv_abs_b)
{
    const T A = fabs(a), Bv = fabs(b);
    T q = trunc(A * inv_abs_b); // a guess within one of trunc(A / Bv): the checks below settle it
    if (!(q < (T)(sizeof(T) == 4 ? 4194304.0 : 2251799813685248.0))) return fmod(a, b);
    T r = fma(-q, Bv, A);
    if (r < T(0)) r = fma(-(q - T(1)), Bv, A);
    else if (r >= Bv) r = fma(-(q + T(1)), Bv, A);
    return copysign(r, a);
}

template <typename T>
__global__ __launch_bounds__(64) void k_unwrap(T* __restrict__ x, size_t len, T divisor)
{
    constexpr int TILE = 2048;
    __shared__ T tin[TILE];
    __shared__ T tout[TILE]; // a second array: the walker's reads never wait for its own writes
    const T half = divisor / T(2);
    const T inv = T(1) / fabs(divisor);
    T prev = T(0);
    for (size_t base = 0; base < len; base += TILE) {
        const int m = len - base < (size_t)TILE ? (int)(len - base) : TILE;
        for (int i = threadIdx.x; i < m; i += 64) tin[i] = x[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            int j = 0;
            if (base == 0) { prev = tin[0]; tout[0] = prev; j = 1; }
#pragma unroll 8
            for (; j < m; ++j) {
                T cur = tin[j];
                T diff = cur - prev;
                if (diff > half) { diff = fmod_exact(diff, divisor, inv); diff = diff - divisor; cur = prev + diff; }
                else if (diff < -half) { diff = fmod_exact(diff, divisor, inv); diff = diff + divisor; cur = prev + diff; }
                tout[j] = cur;
                prev = cur;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 64) x[base + i] = tout[i];
        __syncthreads();
    }
}
template <typename T> int vm_unwrap(T* x, size_t len, T divisor, hipStream_t s)
{
    if (len < 2) return BDSP_OK;
    hipLaunchKernelGGL(k_unwrap<T>, dim3(1), dim3(64), 0, s, x, len, divisor);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- real/imag and magnitude/phase pairs ------------------------------------------------------------------------
// kind 0: (re, im) -> a, b   kind 1: (|z|, arg z) -> a, b   (to_polar = (hypot, atan2))
template <typename T>
__global__ __launch_bounds__(256) void k_complex_split(const T* __restrict__ x, T* __restrict__ a, T* __restrict__ b,
                                                       size_t points, int kind)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < points; i += (size_t)gridDim.x * 256) {
        const T re = x[2 * i], im = x[2 * i + 1];
        if (kind == 0) { a[i] = re; b[i] = im; }
        else { a[i] = hypot(re, im); b[i] = atan2(im, re); }
    }
}
// kind 0: z = (a, b)   kind 1: z = from_polar(a, b)
template <typename T>
__global__ __launch_bounds__(256) void k_complex_join(T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ b,
                                                      size_t points, int kind)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < points; i += (size_t)gridDim.x * 256) {
        if (kind == 0) { x[2 * i] = a[i]; x[2 * i + 1] = b[i]; }
        else { const Cx<T> z = cx_from_polar<T>(a[i], b[i]); x[2 * i] = z.re; x[2 * i + 1] = z.im; }
    }
}
template <typename T> int vm_complex_split(const T* x, T* a, T* b, size_t points, int kind, hipStream_t s)
{
    if (points == 0) return BDSP_OK;
    hipLaunchKernelGGL(k_complex_split<T>, dim3(ew_grid(points)), dim3(256), 0, s, x, a, b, points, kind);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T> int vm_complex_join(T* x, const T* a, const T* b, size_t points, int kind, hipStream_t s)
{
    if (points == 0) return BDSP_OK;
    hipLaunchKernelGGL(k_complex_join<T>, dim3(ew_grid(points)), dim3(256), 0, s, x, a, b, points, kind);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---- split_into / merge: element i <-> part i % n, position i / n; `parts` is a device array of n pointers -----
template <typename T, int E>
__global__ __launch_bounds__(256) void k_split_merge(T* __restrict__ whole, T* const* __restrict__ parts, size_t elements,
                                                     unsigned n, bool merge)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < elements; i += (size_t)gridDim.x * 256) {
        T* part = parts[i % n];
        const size_t pos = i / n;
        for (int c = 0; c < E; ++c) {
            if (merge) whole[i * E + c] = part[pos * E + c];
            else part[pos * E + c] = whole[i * E + c];
        }
    }
}
template <typename T> int vm_split_merge(T* whole, T* const* parts_dev, size_t len, bool is_complex, size_t n, bool merge, hipStream_t s)
{
    const size_t elements = is_complex ? len / 2 : len;
    if (elements == 0) return BDSP_OK;
    if (is_complex) hipLaunchKernelGGL((k_split_merge<T, 2>), dim3(ew_grid(elements)), dim3(256), 0, s, whole, parts_dev, elements, (unsigned)n, merge);
    else hipLaunchKernelGGL((k_split_merge<T, 1>), dim3(ew_grid(elements)), dim3(256), 0, s, whole, parts_dev, elements, (unsigned)n, merge);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

#define BDSP_INST(T)                                                                                   \
    template int ew_math<T>(T*, size_t, bool, int, T, hipStream_t);                                    \
    template int vm_diff<T>(const T*, T*, size_t, size_t, bool, hipStream_t);                          \
    template size_t vm_cum_sum_scratch<T>(size_t, bool);                                               \
    template int vm_cum_sum<T>(T*, size_t, bool, void*, hipStream_t);                                  \
    template int vm_unwrap<T>(T*, size_t, T, hipStream_t);                                             \
    template int vm_complex_split<T>(const T*, T*, T*, size_t, int, hipStream_t);                      \
    template int vm_complex_join<T>(T*, const T*, const T*, size_t, int, hipStream_t);                 \
    template int vm_split_merge<T>(T*, T* const*, size_t, bool, size_t, bool, hipStream_t);
BDSP_INST(float)
BDSP_INST(double)
#undef BDSP_INST

} // namespace bdsp
