// reduce.hip -- statistics, sums and dot products (vector/src/vector_types/general/statistics.rs:181-530,
// dot_products.rs:67-165): one pass over the vector at HBM rate, a second tiny launch folds the
// per-workgroup partials.  Accumulation is in double whatever T is (the reference adds sequentially in T;
// any order is as valid, a more accurate one is welcome), minimum / maximum follow the reference's rules:
// first occurrence wins, complex values are ordered by norm(), NaNs never win a comparison.
#include "bdsp_internal.h"

namespace bdsp {

template <typename T> __device__ __forceinline__ T dev_norm(T a, T b);
template <> __device__ __forceinline__ float dev_norm<float>(float a, float b) { return hypotf(a, b); }
template <> __device__ __forceinline__ double dev_norm<double>(double a, double b) { return hypot(a, b); }

__device__ __forceinline__ void stat_init(StatPartial& p, bool cplx)
{
    p.sr = p.si = p.qr = p.qi = 0.0;
    p.cnt = 0; p.imn = 0; p.imx = 0;
    if (cplx) { // Statistics<Complex>::empty(): min = (inf, inf), max = (0, 0)   (statistics.rs:270-283)
        p.mnr = p.mni = INFINITY; p.mn_key = INFINITY; p.mxr = p.mxi = 0.0; p.mx_key = 0.0;
    } else {    // min = +inf, max = -inf   (:185-196)
        p.mnr = INFINITY; p.mni = 0.0; p.mn_key = INFINITY; p.mxr = -INFINITY; p.mxi = 0.0; p.mx_key = -INFINITY;
    }
}

// fold b into a: the larger (smaller) key wins the maximum (minimum), equal keys keep the earlier element --
// exactly what one sequential walk with strict comparisons produces (statistics.rs:251-263, 341-353)
__device__ __forceinline__ void stat_merge_ordered(StatPartial& a, const StatPartial& b)
{
    a.sr += b.sr; a.si += b.si; a.qr += b.qr; a.qi += b.qi; a.cnt += b.cnt;
    if (b.mx_key > a.mx_key || (b.mx_key == a.mx_key && b.imx < a.imx)) {
        a.mx_key = b.mx_key; a.mxr = b.mxr; a.mxi = b.mxi; a.imx = b.imx;
    }
    if (b.mn_key < a.mn_key || (b.mn_key == a.mn_key && b.imn < a.imn)) {
        a.mn_key = b.mn_key; a.mnr = b.mnr; a.mni = b.mni; a.imn = b.imn;
    }
}

// element j of the walk is x[first + j*step] (statistics_split: first = bucket, step = len), j < count
template <typename T, bool CPLX, bool MINMAX>
__global__ __launch_bounds__(256) void k_stats(const T* __restrict__ x, size_t count, size_t first, size_t step,
                                               StatPartial* __restrict__ partials)
{
    __shared__ StatPartial sh[256];
    StatPartial p;
    stat_init(p, CPLX);
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < count; j += (size_t)gridDim.x * 256) {
        const size_t i = first + j * step;
        if (CPLX) {
            const T re = x[2 * i], im = x[2 * i + 1];
            p.sr += (double)re; p.si += (double)im;
            p.qr += (double)re * (double)re - (double)im * (double)im;
            p.qi += 2.0 * (double)re * (double)im;
            if (MINMAX) {
                const double key = (double)dev_norm<T>(re, im);
                if (key > p.mx_key) { p.mx_key = key; p.mxr = re; p.mxi = im; p.imx = j; }
                if (key < p.mn_key) { p.mn_key = key; p.mnr = re; p.mni = im; p.imn = j; }
            }
        } else {
            const T e = x[i];
            p.sr += (double)e; p.qr += (double)e * (double)e;
            if (MINMAX) {
                if ((double)e > p.mx_key) { p.mx_key = e; p.mxr = e; p.imx = j; }
                if ((double)e < p.mn_key) { p.mn_key = e; p.mnr = e; p.imn = j; }
            }
        }
        p.cnt += 1;
    }
    sh[threadIdx.x] = p;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) stat_merge_ordered(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void k_stats_final(StatPartial* __restrict__ partials, int n, bool cplx)
{
    __shared__ StatPartial sh[256];
    StatPartial p;
    stat_init(p, cplx);
    for (int i = threadIdx.x; i < n; i += 256) stat_merge_ordered(p, partials[i]);
    sh[threadIdx.x] = p;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) stat_merge_ordered(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[0] = sh[0];
}

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_dot(const T* __restrict__ x, const T* __restrict__ y, size_t count,
                                             StatPartial* __restrict__ partials)
{
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < count; j += (size_t)gridDim.x * 256) {
        if (CPLX) {
            const double ar = x[2 * j], ai = x[2 * j + 1], br = y[2 * j], bi = y[2 * j + 1];
            a += ar * br - ai * bi;
            b += ar * bi + ai * br;
        } else {
            a += (double)x[j] * (double)y[j];
        }
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        StatPartial p;
        stat_init(p, CPLX);
        p.sr = sh[0][0]; p.si = sh[1][0];
        partials[blockIdx.x] = p;
    }
}

static unsigned red_grid(size_t count)
{
    size_t g = (count + 255) / 256, cap = (size_t)num_cus() * 4;
    if (cap > 1024) cap = 1024;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// `partials` holds at least 1024 entries; the folded result ends in partials[0]
template <typename T>
int red_stats(const T* x, size_t count, size_t first, size_t step, bool is_complex, bool minmax, StatPartial* partials,
              hipStream_t s)
{
    const unsigned g = red_grid(count);
    if (is_complex) {
        if (minmax) hipLaunchKernelGGL((k_stats<T, true, true>), dim3(g), dim3(256), 0, s, x, count, first, step, partials);
        else hipLaunchKernelGGL((k_stats<T, true, false>), dim3(g), dim3(256), 0, s, x, count, first, step, partials);
    } else {
        if (minmax) hipLaunchKernelGGL((k_stats<T, false, true>), dim3(g), dim3(256), 0, s, x, count, first, step, partials);
        else hipLaunchKernelGGL((k_stats<T, false, false>), dim3(g), dim3(256), 0, s, x, count, first, step, partials);
    }
    BDSP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(256), 0, s, partials, (int)g, is_complex);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template <typename T>
int red_dot(const T* x, const T* y, size_t count, bool is_complex, StatPartial* partials, hipStream_t s)
{
    const unsigned g = red_grid(count);
    if (is_complex) hipLaunchKernelGGL((k_dot<T, true>), dim3(g), dim3(256), 0, s, x, y, count, partials);
    else hipLaunchKernelGGL((k_dot<T, false>), dim3(g), dim3(256), 0, s, x, y, count, partials);
    BDSP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(256), 0, s, partials, (int)g, is_complex);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template int red_stats<float>(const float*, size_t, size_t, size_t, bool, bool, StatPartial*, hipStream_t);
template int red_stats<double>(const double*, size_t, size_t, size_t, bool, bool, StatPartial*, hipStream_t);
template int red_dot<float>(const float*, const float*, size_t, bool, StatPartial*, hipStream_t);
template int red_dot<double>(const double*, const double*, size_t, bool, StatPartial*, hipStream_t);

} // namespace bdsp
