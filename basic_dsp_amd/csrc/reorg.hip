// reorg.hip -- pure index moves (bit-exact): swap_halves / fft_shift / ifft_shift, reverse,
// zero_pad, zero_interleave, mirror.  All out-of-place (the handle trades buffers afterwards, like
// the reference's Buffer::trade, vector/src/vector_types/support_std.rs:78-82).
//   swap_array_halves   vector/src/vector_types/mod.rs:171-191
//   reverse / zero_pad / zero_interleave   vector/src/vector_types/general/data_reorganization.rs:237-479
//   mirror              vector/src/vector_types/time_freq/freq.rs:52-83
#include "bdsp_internal.h"

namespace bdsp {

static inline unsigned rg_grid(size_t n)
{
    size_t blocks = (n + 255) / 256;
    size_t cap = (size_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks ? blocks : 1);
}

// out[i] = in[(i + shift) mod points], on elements of `elem` scalars.
// fft_shift: shift = ceil(points/2); ifft_shift: shift = floor(points/2) -- for odd lengths this is
// exactly what the reference's cycle walk produces (KATs vector_types/mod.rs:700-712).
template <typename T>
__global__ __launch_bounds__(256) void k_rotate(const T* __restrict__ in, T* __restrict__ out,
                                                 size_t points, size_t elem, size_t shift)
{
    size_t total = points * elem;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        size_t i = g / elem, e = g % elem;
        size_t src = i + shift;
        if (src >= points) src -= points;
        out[g] = in[src * elem + e];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_reverse(const T* __restrict__ in, T* __restrict__ out,
                                                  size_t points, size_t elem)
{
    size_t total = points * elem;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        size_t i = g / elem, e = g % elem;
        out[g] = in[(points - 1 - i) * elem + e];
    }
}

// out = zeros(len); out[dst0 .. dst0+n0) = in[src0 ..); out[dst1 .. dst1+n1) = in[src1 ..)
template <typename T>
__global__ __launch_bounds__(256) void k_two_segment_copy(const T* __restrict__ in, T* __restrict__ out,
                                                           size_t len, size_t dst0, size_t src0, size_t n0,
                                                           size_t dst1, size_t src1, size_t n1)
{
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < len;
         g += (size_t)gridDim.x * blockDim.x) {
        T v = (T)0;
        if (g >= dst0 && g < dst0 + n0) v = in[src0 + (g - dst0)];
        else if (g >= dst1 && g < dst1 + n1) v = in[src1 + (g - dst1)];
        out[g] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_zero_interleave(const T* __restrict__ in, T* __restrict__ out,
                                                          size_t points, size_t elem, size_t factor)
{
    size_t total = points * factor * elem;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        size_t i = g / elem, e = g % elem;
        out[g] = (i % factor == 0) ? in[(i / factor) * elem + e] : (T)0;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_mirror(const cpx<T>* __restrict__ in, cpx<T>* __restrict__ out,
                                                 size_t p)
{
    size_t total = 2 * p - 1;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        if (g < p) out[g] = in[g];
        else {
            cpx<T> z = in[2 * p - 1 - g]; // g = p-1+k  ->  in[p-k]
            out[g] = cpx<T>{z.x, -z.y};
        }
    }
}

// decimatei (interpolation.rs:607-633): out[j] = in[delay + j*factor]
template <typename T>
__global__ __launch_bounds__(256) void k_decimate(const T* __restrict__ in, T* __restrict__ out, size_t out_points,
                                                   size_t elem, size_t factor, size_t delay)
{
    size_t total = out_points * elem;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        size_t j = g / elem, e = g % elem;
        out[g] = in[(delay + j * factor) * elem + e];
    }
}
template <typename T> int rg_decimate(const T* in, T* out, size_t out_points, size_t elem, size_t factor, size_t delay, hipStream_t s)
{
    if (out_points == 0) return BDSP_OK;
    hipLaunchKernelGGL((k_decimate<T>), dim3(rg_grid(out_points * elem)), dim3(256), 0, s, in, out, out_points, elem, factor, delay);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// out[i] = in[(start + i) mod points], i < total (total may exceed points): the circular extension the
// long-filter overlap-save path transforms its overlapping windows from
template <typename T>
__global__ __launch_bounds__(256) void k_wrap_copy(const T* __restrict__ in, T* __restrict__ out, size_t points,
                                                    size_t elem, size_t total, size_t start)
{
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total * elem;
         g += (size_t)gridDim.x * blockDim.x) {
        size_t i = g / elem, e = g % elem;
        out[g] = in[((start + i) % points) * elem + e];
    }
}
template <typename T> int rg_wrap_copy(const T* in, T* out, size_t points, size_t elem, size_t total, long long start, hipStream_t s)
{
    if (points == 0 || total == 0) return BDSP_OK;
    long long st = start % (long long)points;
    if (st < 0) st += (long long)points;
    hipLaunchKernelGGL((k_wrap_copy<T>), dim3(rg_grid(total * elem)), dim3(256), 0, s, in, out, points, elem, total, (size_t)st);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T> int rg_rotate(const T* in, T* out, size_t points, size_t elem, size_t shift, hipStream_t s)
{
    if (points == 0) return BDSP_OK;
    hipLaunchKernelGGL((k_rotate<T>), dim3(rg_grid(points * elem)), dim3(256), 0, s, in, out, points, elem, shift % points);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T> int rg_reverse(const T* in, T* out, size_t points, size_t elem, hipStream_t s)
{
    if (points == 0) return BDSP_OK;
    hipLaunchKernelGGL((k_reverse<T>), dim3(rg_grid(points * elem)), dim3(256), 0, s, in, out, points, elem);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
// option: 0 End, 1 Surround (zero_pad_b flavour: right = diff/2, data_reorganization.rs:429-442),
// 2 Center (first ceil(P/2) points stay, last floor(P/2) move to the end, :343-358).
template <typename T> int rg_zero_pad(const T* in, T* out, size_t len_before, bool is_complex, size_t points, int option, hipStream_t s)
{
    size_t step = is_complex ? 2 : 1, len = points * step;
    if (len <= len_before) return BDSP_ERR_ARG_LENGTH;
    size_t d0 = 0, s0 = 0, n0 = len_before, d1 = 0, s1 = 0, n1 = 0;
    if (option == 1) {
        size_t diff = (len - len_before) / step, right = diff / 2;
        d0 = (diff - right) * step;
    } else if (option != 0) {
        size_t pb = len_before / step, right = (pb / 2) * step, left = (pb - pb / 2) * step;
        n0 = left;
        d1 = len - right; s1 = len_before - right; n1 = right;
    }
    hipLaunchKernelGGL((k_two_segment_copy<T>), dim3(rg_grid(len)), dim3(256), 0, s, in, out, len, d0, s0, n0, d1, s1, n1);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T> int rg_zero_interleave(const T* in, T* out, size_t len, size_t elem, size_t factor, hipStream_t s)
{
    if (len == 0) return BDSP_OK;
    hipLaunchKernelGGL((k_zero_interleave<T>), dim3(rg_grid(len * factor)), dim3(256), 0, s, in, out, len / elem, elem, factor);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T> int rg_mirror(const T* in, T* out, size_t len, hipStream_t s)
{
    size_t p = len / 2;
    if (p == 0) return BDSP_OK;
    hipLaunchKernelGGL((k_mirror<T>), dim3(rg_grid(2 * p)), dim3(256), 0, s,
                       reinterpret_cast<const cpx<T>*>(in), reinterpret_cast<cpx<T>*>(out), p);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

#define BDSP_INST(T)                                                                               \
    template int rg_rotate<T>(const T*, T*, size_t, size_t, size_t, hipStream_t);                  \
    template int rg_wrap_copy<T>(const T*, T*, size_t, size_t, size_t, long long, hipStream_t);    \
    template int rg_reverse<T>(const T*, T*, size_t, size_t, hipStream_t);                         \
    template int rg_zero_pad<T>(const T*, T*, size_t, bool, size_t, int, hipStream_t);             \
    template int rg_zero_interleave<T>(const T*, T*, size_t, size_t, size_t, hipStream_t);         \
    template int rg_mirror<T>(const T*, T*, size_t, hipStream_t);                                  \
    template int rg_decimate<T>(const T*, T*, size_t, size_t, size_t, size_t, hipStream_t);
BDSP_INST(float)
BDSP_INST(double)
#undef BDSP_INST

} // namespace bdsp
