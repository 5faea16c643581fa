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

template <typename T, bool CPLX, bool MINMAX>
__device__ __forceinline__ void stat_take(StatPartial& p, T re, T im, size_t j)
{
    if (CPLX) {
        p.sr += (double)re; p.si += (double)im;
        p.qr += (double)re * (double)re - (double)im * (double)im;
        p.qi += 2.0 * (double)re * (double)im;
        if (MINMAX) {
            const double key = (double)dev_norm<T>(re, im);
            if (key > p.mx_key) { p.mx_key = key; p.mxr = re; p.mxi = im; p.imx = j; }
            if (key < p.mn_key) { p.mn_key = key; p.mnr = re; p.mni = im; p.imn = j; }
        }
    } else {
        p.sr += (double)re; p.qr += (double)re * (double)re;
        if (MINMAX) {
            if ((double)re > p.mx_key) { p.mx_key = re; p.mxr = re; p.imx = j; }
            if ((double)re < p.mn_key) { p.mn_key = re; p.mnr = re; p.imn = j; }
        }
    }
    p.cnt += 1;
}

__device__ __forceinline__ void stat_block_fold(StatPartial& p, StatPartial* sh, StatPartial* out)
{
    sh[threadIdx.x] = p;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) stat_merge_ordered(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

// The whole vector (first = 0, step = 1): every workgroup owns one contiguous run of 16-byte packets and keeps
// four of them in flight per lane (far-apart concurrent streams thrash DRAM pages, see elementwise.hip).
template <typename T, bool CPLX, bool MINMAX>
__global__ __launch_bounds__(256) void k_stats_contig(const T* __restrict__ x, size_t count,
                                                      StatPartial* __restrict__ partials)
{
    constexpr int TPP = 16 / sizeof(T);            // scalars per packet
    constexpr int EPP = TPP / (CPLX ? 2 : 1);      // elements per packet
    struct alignas(16) Pk { T v[TPP]; };
    __shared__ StatPartial sh[256];
    StatPartial p;
    stat_init(p, CPLX);
    const size_t npk = count / EPP;
    size_t per = (npk + gridDim.x - 1) / gridDim.x;
    per = (per + 1023) / 1024 * 1024;
    const size_t p0 = (size_t)blockIdx.x * per;
    const size_t p1 = p0 + per < npk ? p0 + per : npk;
    const Pk* __restrict__ xp = reinterpret_cast<const Pk*>(x);
    for (size_t q = p0 + threadIdx.x; q < p1; q += 1024) {
        Pk pk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q + 256 * u < p1) pk[u] = xp[q + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (q + 256 * u < p1) {
#pragma unroll
                for (int k = 0; k < EPP; ++k)
                    stat_take<T, CPLX, MINMAX>(p, pk[u].v[CPLX ? 2 * k : k], CPLX ? pk[u].v[2 * k + 1] : T(0),
                                               (q + 256 * u) * EPP + k);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) // the elements past the last whole packet
        for (size_t j = npk * EPP; j < count; ++j)
            stat_take<T, CPLX, MINMAX>(p, CPLX ? x[2 * j] : x[j], CPLX ? x[2 * j + 1] : T(0), j);
    stat_block_fold(p, sh, &partials[blockIdx.x]);
}

// statistics_split: bucket b = blockIdx.y walks x[b + j*nb], j < ceil((total - b) / nb); the buckets run side by
// side so the cache lines they share are fetched from HBM once.  Partials of bucket b live at partials[b*1024 ...].
template <typename T, bool CPLX, bool MINMAX>
__global__ __launch_bounds__(256) void k_stats_strided(const T* __restrict__ x, size_t total,
                                                       StatPartial* __restrict__ partials)
{
    __shared__ StatPartial sh[256];
    StatPartial p;
    stat_init(p, CPLX);
    const size_t first = blockIdx.y, step = gridDim.y;
    const size_t count = total > first ? (total - first + step - 1) / step : 0;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < count; j += (size_t)gridDim.x * 256) {
        const size_t i = first + j * step;
        stat_take<T, CPLX, MINMAX>(p, CPLX ? x[2 * i] : x[i], CPLX ? x[2 * i + 1] : T(0), j);
    }
    stat_block_fold(p, sh, &partials[(size_t)blockIdx.y * 1024 + blockIdx.x]);
}

// one workgroup per bucket folds `n` partials at partials[bucket*1024 ...] into out[bucket] (pinned host memory:
// the result needs no copy, only the stream synchronisation)
__global__ __launch_bounds__(256) void k_stats_final(StatPartial* __restrict__ partials, int n, bool cplx,
                                                     StatPartial* __restrict__ out)
{
    __shared__ StatPartial sh[256];
    StatPartial p;
    stat_init(p, cplx);
    partials += (size_t)blockIdx.x * 1024;
    for (int i = threadIdx.x; i < n; i += 256) stat_merge_ordered(p, partials[i]);
    __syncthreads();
    stat_block_fold(p, sh, &out[blockIdx.x]);
}

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_dot(const T* __restrict__ x, const T* __restrict__ y, size_t count,
                                             StatPartial* __restrict__ partials)
{
    constexpr int TPP = 16 / sizeof(T);
    constexpr int EPP = TPP / (CPLX ? 2 : 1);
    struct alignas(16) Pk { T v[TPP]; };
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    const size_t npk = count / EPP;
    size_t per = (npk + gridDim.x - 1) / gridDim.x;
    per = (per + 511) / 512 * 512;
    const size_t p0 = (size_t)blockIdx.x * per;
    const size_t p1 = p0 + per < npk ? p0 + per : npk;
    const Pk* __restrict__ xp = reinterpret_cast<const Pk*>(x);
    const Pk* __restrict__ yp = reinterpret_cast<const Pk*>(y);
    auto take = [&](T ar, T ai, T br, T bi) {
        if (CPLX) {
            a += (double)ar * (double)br - (double)ai * (double)bi;
            b += (double)ar * (double)bi + (double)ai * (double)br;
        } else {
            a += (double)ar * (double)br;
        }
    };
    for (size_t q = p0 + threadIdx.x; q < p1; q += 512) {
        Pk px[2], py[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (q + 256 * u < p1) { px[u] = xp[q + 256 * u]; py[u] = yp[q + 256 * u]; }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (q + 256 * u < p1) {
#pragma unroll
                for (int k = 0; k < EPP; ++k)
                    take(px[u].v[CPLX ? 2 * k : k], CPLX ? px[u].v[2 * k + 1] : T(0), py[u].v[CPLX ? 2 * k : k],
                         CPLX ? py[u].v[2 * k + 1] : T(0));
            }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t j = npk * EPP; j < count; ++j)
            take(CPLX ? x[2 * j] : x[j], CPLX ? x[2 * j + 1] : T(0), CPLX ? y[2 * j] : y[j], CPLX ? y[2 * j + 1] : T(0));
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

// `partials` holds at least 1024 * buckets entries; bucket b's folded result ends in out[b] (device-visible).
// buckets == 1 is the plain statistics of the whole vector.
template <typename T>
int red_stats(const T* x, size_t total, size_t buckets, bool is_complex, bool minmax, StatPartial* partials,
              StatPartial* out, hipStream_t s)
{
    if (buckets <= 1) {
        const size_t epp = 16 / sizeof(T) / (is_complex ? 2 : 1);
        const unsigned g = red_grid(total / epp / 4);
        if (is_complex) {
            if (minmax) hipLaunchKernelGGL((k_stats_contig<T, true, true>), dim3(g), dim3(256), 0, s, x, total, partials);
            else hipLaunchKernelGGL((k_stats_contig<T, true, false>), dim3(g), dim3(256), 0, s, x, total, partials);
        } else {
            if (minmax) hipLaunchKernelGGL((k_stats_contig<T, false, true>), dim3(g), dim3(256), 0, s, x, total, partials);
            else hipLaunchKernelGGL((k_stats_contig<T, false, false>), dim3(g), dim3(256), 0, s, x, total, partials);
        }
        BDSP_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(256), 0, s, partials, (int)g, is_complex, out);
        BDSP_LAUNCH_CHECK();
        return BDSP_OK;
    }
    const unsigned g = red_grid((total + buckets - 1) / buckets);
    const dim3 grid(g, (unsigned)buckets);
    if (is_complex) hipLaunchKernelGGL((k_stats_strided<T, true, true>), grid, dim3(256), 0, s, x, total, partials);
    else hipLaunchKernelGGL((k_stats_strided<T, false, true>), grid, dim3(256), 0, s, x, total, partials);
    BDSP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_stats_final, dim3((unsigned)buckets), dim3(256), 0, s, partials, (int)g, is_complex, out);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template <typename T>
int red_dot(const T* x, const T* y, size_t count, bool is_complex, StatPartial* partials, StatPartial* out,
            hipStream_t s)
{
    const unsigned g = red_grid(count / (16 / sizeof(T) / (is_complex ? 2 : 1)) / 2);
    if (is_complex) hipLaunchKernelGGL((k_dot<T, true>), dim3(g), dim3(256), 0, s, x, y, count, partials);
    else hipLaunchKernelGGL((k_dot<T, false>), dim3(g), dim3(256), 0, s, x, y, count, partials);
    BDSP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(256), 0, s, partials, (int)g, is_complex, out);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template int red_stats<float>(const float*, size_t, size_t, bool, bool, StatPartial*, StatPartial*, hipStream_t);
template int red_stats<double>(const double*, size_t, size_t, bool, bool, StatPartial*, StatPartial*, hipStream_t);
template int red_dot<float>(const float*, const float*, size_t, bool, StatPartial*, StatPartial*, hipStream_t);
template int red_dot<double>(const double*, const double*, size_t, bool, StatPartial*, StatPartial*, hipStream_t);

} // namespace bdsp
