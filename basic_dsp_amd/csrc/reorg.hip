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

// Every kernel moves whole ELEMENTS (a real scalar or an interleaved complex pair) as one packet P of
// 4..16 bytes, so a lane issues one load and one store per element and no index is divided by `elem`;
// indices are 32-bit whenever the vector allows it (IDX).  rg_dispatch picks P and IDX.
template <typename T, int ELEM> struct packet_of { using type = T; };
template <typename T> struct packet_of<T, 2> { using type = cpx<T>; };

template <typename P> __device__ __forceinline__ P zero_packet() { P z; __builtin_memset(&z, 0, sizeof(P)); return z; }

// out[i] = in[(i + shift) mod points].
// fft_shift: shift = ceil(points/2); ifft_shift: shift = floor(points/2) -- for odd lengths this is
// exactly what the reference's cycle walk produces (KATs vector_types/mod.rs:700-712).
template <typename P, typename IDX>
__global__ __launch_bounds__(256) void k_rotate(const P* __restrict__ in, P* __restrict__ out, IDX points, IDX shift)
{
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < points; i += (IDX)gridDim.x * blockDim.x) {
        IDX src = i + shift;
        if (src >= points) src -= points;
        out[i] = in[src];
    }
}

template <typename P, typename IDX>
__global__ __launch_bounds__(256) void k_reverse(const P* __restrict__ in, P* __restrict__ out, IDX points)
{
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < points; i += (IDX)gridDim.x * blockDim.x)
        out[i] = in[points - 1 - i];
}

// out = zeros(len); out[dst0 .. dst0+n0) = in[src0 ..); out[dst1 .. dst1+n1) = in[src1 ..)   (in elements)
template <typename P, typename IDX>
__global__ __launch_bounds__(256) void k_two_segment_copy(const P* __restrict__ in, P* __restrict__ out, IDX len,
                                                           IDX dst0, IDX src0, IDX n0, IDX dst1, IDX src1, IDX n1)
{
    for (IDX g = (IDX)blockIdx.x * blockDim.x + threadIdx.x; g < len; g += (IDX)gridDim.x * blockDim.x) {
        P v = zero_packet<P>();
        if (g >= dst0 && g - dst0 < n0) v = in[src0 + (g - dst0)];
        else if (g >= dst1 && g - dst1 < n1) v = in[src1 + (g - dst1)];
        out[g] = v;
    }
}

// out[i*factor] = in[i], zero elsewhere (one thread per OUTPUT element: contiguous stores)
template <typename P, typename IDX>
__global__ __launch_bounds__(256) void k_zero_interleave(const P* __restrict__ in, P* __restrict__ out, IDX points,
                                                          IDX factor)
{
    const IDX total = points * factor;
    for (IDX g = (IDX)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (IDX)gridDim.x * blockDim.x) {
        const IDX q = g / factor;
        out[g] = (g - q * factor == 0) ? in[q] : zero_packet<P>();
    }
}
// factor 2 (to_complex, interpolation by two): one 2-element store per input element
template <typename P, typename IDX>
__global__ __launch_bounds__(256) void k_zero_interleave2(const P* __restrict__ in, P* __restrict__ out, IDX points)
{
    struct alignas(2 * sizeof(P)) P2 { P a, b; };
    P2* o2 = reinterpret_cast<P2*>(out);
    if constexpr (sizeof(P) == 4) {
        // 4-byte elements (real f32 -> to_complex): two per lane, one 8-byte load and one 16-byte store
        struct alignas(16) P4 { P a, z0, b, z1; };
        const P2* i2 = reinterpret_cast<const P2*>(in);
        P4* o4 = reinterpret_cast<P4*>(out);
        const IDX pairs = points / 2;
        for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (IDX)gridDim.x * blockDim.x) {
            const P2 v = i2[i];
            o4[i] = P4{v.a, zero_packet<P>(), v.b, zero_packet<P>()};
        }
        if ((points & 1) && blockIdx.x == 0 && threadIdx.x == 0) o2[points - 1] = P2{in[points - 1], zero_packet<P>()};
    } else {
        for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < points; i += (IDX)gridDim.x * blockDim.x)
            o2[i] = P2{in[i], zero_packet<P>()};
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
template <typename P, typename IDX>
__global__ __launch_bounds__(256) void k_decimate(const P* __restrict__ in, P* __restrict__ out, IDX out_points,
                                                   IDX factor, IDX delay)
{
    for (IDX j = (IDX)blockIdx.x * blockDim.x + threadIdx.x; j < out_points; j += (IDX)gridDim.x * blockDim.x)
        out[j] = in[delay + j * factor];
}

// out[i] = in[(start + i) mod points], i < total (total may exceed points): the circular extension the
// long-filter overlap-save path transforms its overlapping windows from
template <typename P, typename IDX>
__global__ __launch_bounds__(256) void k_wrap_copy(const P* __restrict__ in, P* __restrict__ out, IDX points, IDX total,
                                                    IDX start)
{
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x)
        out[i] = in[(start + i) % points];
}

// launch KERNEL<P, IDX>(args...) with P = the element packet and IDX = 32-bit indices when `span` allows
#define BDSP_RG_LAUNCH(KERNEL, span, ...)                                                                      \
    do {                                                                                                       \
        const size_t span_ = (span);                                                                           \
        if (elem == 2) {                                                                                       \
            using P = cpx<T>;                                                                                  \
            if (span_ < (size_t(1) << 31)) hipLaunchKernelGGL((KERNEL<P, unsigned>), dim3(rg_grid(span_)), dim3(256), 0, s, BDSP_RG_ARGS(P, unsigned)); \
            else hipLaunchKernelGGL((KERNEL<P, size_t>), dim3(rg_grid(span_)), dim3(256), 0, s, BDSP_RG_ARGS(P, size_t)); \
        } else {                                                                                               \
            using P = T;                                                                                       \
            if (span_ < (size_t(1) << 31)) hipLaunchKernelGGL((KERNEL<P, unsigned>), dim3(rg_grid(span_)), dim3(256), 0, s, BDSP_RG_ARGS(P, unsigned)); \
            else hipLaunchKernelGGL((KERNEL<P, size_t>), dim3(rg_grid(span_)), dim3(256), 0, s, BDSP_RG_ARGS(P, size_t)); \
        }                                                                                                      \
        BDSP_LAUNCH_CHECK();                                                                                   \
    } while (0)

template <typename T> int rg_decimate(const T* in, T* out, size_t out_points, size_t elem, size_t factor, size_t delay, hipStream_t s)
{
    if (out_points == 0) return BDSP_OK;
#define BDSP_RG_ARGS(P, I) reinterpret_cast<const P*>(in), reinterpret_cast<P*>(out), (I)out_points, (I)factor, (I)delay
    BDSP_RG_LAUNCH(k_decimate, (delay + out_points * factor) > out_points ? delay + out_points * factor : out_points);
#undef BDSP_RG_ARGS
    return BDSP_OK;
}

template <typename T> int rg_wrap_copy(const T* in, T* out, size_t points, size_t elem, size_t total, long long start, hipStream_t s)
{
    if (points == 0 || total == 0) return BDSP_OK;
    long long st = start % (long long)points;
    if (st < 0) st += (long long)points;
#define BDSP_RG_ARGS(P, I) reinterpret_cast<const P*>(in), reinterpret_cast<P*>(out), (I)points, (I)total, (I)st
    BDSP_RG_LAUNCH(k_wrap_copy, total + points);
#undef BDSP_RG_ARGS
    return BDSP_OK;
}
template <typename T> int rg_rotate(const T* in, T* out, size_t points, size_t elem, size_t shift, hipStream_t s)
{
    if (points == 0) return BDSP_OK;
    const size_t sh = shift % points;
#define BDSP_RG_ARGS(P, I) reinterpret_cast<const P*>(in), reinterpret_cast<P*>(out), (I)points, (I)sh
    BDSP_RG_LAUNCH(k_rotate, 2 * points);
#undef BDSP_RG_ARGS
    return BDSP_OK;
}
template <typename T> int rg_reverse(const T* in, T* out, size_t points, size_t elem, hipStream_t s)
{
    if (points == 0) return BDSP_OK;
#define BDSP_RG_ARGS(P, I) reinterpret_cast<const P*>(in), reinterpret_cast<P*>(out), (I)points
    BDSP_RG_LAUNCH(k_reverse, points);
#undef BDSP_RG_ARGS
    return BDSP_OK;
}
// option: 0 End, 1 Surround (zero_pad_b flavour: right = diff/2, data_reorganization.rs:429-442),
// 2 Center (first ceil(P/2) points stay, last floor(P/2) move to the end, :343-358).
template <typename T> int rg_zero_pad(const T* in, T* out, size_t len_before, bool is_complex, size_t points, int option, hipStream_t s)
{
    const size_t elem = is_complex ? 2 : 1, len = points * elem;
    if (len <= len_before) return BDSP_ERR_ARG_LENGTH;
    const size_t pb = len_before / elem; // all offsets below are in ELEMENTS
    size_t d0 = 0, s0 = 0, n0 = pb, d1 = 0, s1 = 0, n1 = 0;
    if (option == 1) {
        size_t diff = points - pb, right = diff / 2;
        d0 = diff - right;
    } else if (option != 0) {
        size_t right = pb / 2, left = pb - pb / 2;
        n0 = left;
        d1 = points - right; s1 = pb - right; n1 = right;
    }
#define BDSP_RG_ARGS(P, I) reinterpret_cast<const P*>(in), reinterpret_cast<P*>(out), (I)points, (I)d0, (I)s0, (I)n0, (I)d1, (I)s1, (I)n1
    BDSP_RG_LAUNCH(k_two_segment_copy, points);
#undef BDSP_RG_ARGS
    return BDSP_OK;
}
template <typename T> int rg_zero_interleave(const T* in, T* out, size_t len, size_t elem, size_t factor, hipStream_t s)
{
    if (len == 0) return BDSP_OK;
    const size_t points = len / elem;
    if (factor == 2) {
#define BDSP_RG_ARGS(P, I) reinterpret_cast<const P*>(in), reinterpret_cast<P*>(out), (I)points
        BDSP_RG_LAUNCH(k_zero_interleave2, points);
#undef BDSP_RG_ARGS
        return BDSP_OK;
    }
#define BDSP_RG_ARGS(P, I) reinterpret_cast<const P*>(in), reinterpret_cast<P*>(out), (I)points, (I)factor
    BDSP_RG_LAUNCH(k_zero_interleave, points * factor);
#undef BDSP_RG_ARGS
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
