// bluestein.hip -- chirp-z kernels for FFT lengths that are not powers of two (the reference takes
// any N through rustfft, time_freq/mod.rs:47-58):
//     X[k] = conj(c[k]) * sum_n (x[n] conj(c[n])) c[k-n],      c[n] = exp(-+ i*pi*n^2/N)
// evaluated as a circular convolution of length m = next_pow2(2N-1) on the power-of-two kernels.
// Four small kernels around two (batched) power-of-two transforms:
//     chirp   c[i] for i < N, phase from the EXACT integer i^2 mod 2N (sincospi in double)
//     b       b[i] = c[i], b[m-i] = c[i], zero elsewhere  (its spectrum is cached per (N, direction))
//     pre     a[v][i] = window(i) * scale * x[v][(i + rot) mod N] * conj(c[i]), zero padded to m
//     post    out[v][i] = conv[v][j] * conj(c[j]), j = (i + rot) mod N, optionally / window(i),
//             as complex, real part or magnitude
// so the prologue/epilogue options of the fused power-of-two path (shift, window, 1/N scale, real
// input, magnitude) cost no extra pass here either.
#include "bdsp_internal.h"
#include "dsp_funcs.h"
#include <type_traits>

namespace bdsp {

template <typename T>
__global__ __launch_bounds__(256) void k_bs_chirp(cpx<T>* __restrict__ c, unsigned long long n, int inverse)
{
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // i < 2^30 so i*i < 2^60: exact in 64 bits
    unsigned long long r = (i * i) % (2 * n);
    double sn, cs;
    sincospi((double)r / (double)n, &sn, &cs);
    // forward transform: c[n] = exp(+i*pi*n^2/N) is the convolution kernel, data are multiplied by
    // its conjugate; the inverse transform conjugates everything
    c[i] = cpx<T>{(T)cs, (T)(inverse ? -sn : sn)};
}

template <typename T>
__global__ __launch_bounds__(256) void k_bs_kernel(const cpx<T>* __restrict__ c, cpx<T>* __restrict__ b,
                                                   unsigned long long n, unsigned long long m)
{
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    cpx<T> v{(T)0, (T)0};
    if (i < n) v = c[i];
    else if (m - i < n) v = c[m - i];
    b[i] = v;
}

template <typename T, bool IN_REAL>
__global__ __launch_bounds__(256) void k_bs_pre(const T* __restrict__ x, cpx<T>* __restrict__ a,
                                                const cpx<T>* __restrict__ c, unsigned long long n,
                                                unsigned long long m, T in_scale, unsigned long long rot,
                                                int window_id, T alpha)
{
    const unsigned long long vec = blockIdx.y;
    const T* xv = x + vec * n * (IN_REAL ? 1 : 2);
    cpx<T>* av = a + vec * m;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < m;
         i += (unsigned long long)gridDim.x * 256) {
        cpx<T> o{(T)0, (T)0};
        if (i < n) {
            unsigned long long j = i + rot;
            if (j >= n) j -= n;
            T re = IN_REAL ? xv[j] : xv[2 * j];
            T im = IN_REAL ? (T)0 : xv[2 * j + 1];
            T w = in_scale;
            if (window_id >= 0) w = w * window_value_sym<T>(window_id, alpha, (size_t)i, (size_t)n);
            re = re * w;
            im = im * w;
            cpx<T> cc = c[i]; // multiply by conj(c)
            o = cpx<T>{re * cc.x + im * cc.y, im * cc.x - re * cc.y};
        }
        av[i] = o;
    }
}

// OUT: 0 complex, 1 real part, 2 magnitude
template <typename T, int OUT>
__global__ __launch_bounds__(256) void k_bs_post(const cpx<T>* __restrict__ conv, T* __restrict__ out,
                                                 const cpx<T>* __restrict__ c, unsigned long long n,
                                                 unsigned long long m, unsigned long long rot,
                                                 int div_window_id, T alpha)
{
    const unsigned long long vec = blockIdx.y;
    const cpx<T>* cv = conv + vec * m;
    T* ov = out + vec * n * (OUT == 0 ? 2 : 1);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        unsigned long long j = i + rot;
        if (j >= n) j -= n;
        cpx<T> z = cv[j], cc = c[j];
        T re = z.x * cc.x + z.y * cc.y, im = z.y * cc.x - z.x * cc.y;
        if (div_window_id >= 0) {
            T w = window_value_sym<T>(div_window_id, alpha, (size_t)i, (size_t)n);
            re = re / w;
            im = im / w;
        }
        if (OUT == 0) { ov[2 * i] = re; ov[2 * i + 1] = im; }
        else if (OUT == 1) ov[i] = re;
        else ov[i] = sizeof(T) == 4 ? (T)hypotf((float)re, (float)im) : (T)hypot((double)re, (double)im);
    }
}

// Whole chirp-z transform in ONE workgroup-resident kernel for m <= 4096 (n <= 2048), plain complex I/O:
//   load x[i] conj(c[i]) (zero padded) -> FFT_m -> x B/m -> IFFT_m -> X[k] = z[k] conj(c[k]), k < n
// exactly the shape of the fused overlap-save block.  The forward transform leaves the spectrum in the
// registers the inverse transform's first stage reads (up to a compile-time renaming), so nothing but the
// four stage exchanges touches LDS.  256/(m/16) transforms per workgroup.
template <typename T, int M>
__global__ __launch_bounds__(256) void k_bluestein_wg(const cpx<T>* __restrict__ x, cpx<T>* __restrict__ y,
                                                      const cpx<T>* __restrict__ c, const cpx<T>* __restrict__ bspec,
                                                      const cpx<T>* __restrict__ wtab, unsigned n, size_t batch)
{
    constexpr int NT = M / 16, B = 256 / NT;
    using F = WgFft<T, M, NT>;
    using P = Radix16Plan<M>;
    constexpr int R2 = P::R2, R3 = P::R3;
    static_assert(R2 == 16 && R3 >= 2, "512 <= M <= 4096");
    constexpr int NSL = M / R3, NTW3 = (16 / R3) * (R3 - 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, col = tid / NT, t = tid % NT;
    cpx<T>* l = reinterpret_cast<cpx<T>*>(smem_raw) + (size_t)col * (F::LDS_ELEMS + 1);
    cpx<T>* tw2l = reinterpret_cast<cpx<T>*>(smem_raw) + (size_t)B * (F::LDS_ELEMS + 1);
    auto tw = [&](int mm) { return wtab[mm]; };
    const T inv_m = (T)1 / (T)M;
    // persistent over the batch: last-stage twiddles in registers, second-stage twiddles in an LDS table
    cpx<T> tw3[NTW3];
    F::template load_twiddles<R3, 256>(tw3, t, tw);
    if (tid < 240) {
        int k = tid / 15, r = tid % 15 + 1;
        tw2l[k * 17 + r - 1] = wtab[r * k * (M / 256)];
    }
    __syncthreads();
    const cpx<T>* tw2p = tw2l + (t & 15) * 17;

    cpx<T> v[16];
    auto fft3 = [&](auto dir) {
        constexpr int DIR = decltype(dir)::value;
        F::template compute<16, 1, DIR>(v, t, tw);
        __syncthreads(); // the previous exchange's gather is done
        F::template scatter<16, 1>(v, t, l);
        __syncthreads();
        F::template gather<16>(v, t, l);
        F::template compute_pre<16, 16, DIR>(v, tw2p);
        __syncthreads();
        F::template scatter<16, 16>(v, t, l);
        __syncthreads();
        F::template gather<R3>(v, t, l);
        F::template compute_pre<R3, 256, DIR>(v, tw3);
    };
    const size_t groups = (batch + B - 1) / B;
    for (size_t gi = blockIdx.x; gi < groups; gi += gridDim.x) {
        const size_t vec = gi * B + col;
        const bool active = vec < batch;
        const cpx<T>* xv = x + vec * (size_t)n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned idx = (unsigned)(t + r * NT);
            cpx<T> a{(T)0, (T)0};
            if (active && idx < n) {
                const cpx<T> xx = xv[idx], cc = c[idx];
                a = cpx<T>{xx.x * cc.x + xx.y * cc.y, xx.y * cc.x - xx.x * cc.y};
            }
            v[r] = a;
        }
        fft3(std::integral_constant<int, -1>{});
        // register b*R3 + r holds spectrum index k = t + b*NT + r*NSL = t + NT*(b + r*(16/R3)); the inverse
        // transform's first stage wants index t + NT*r' in register r': rename, and multiply by B/m on the way
        cpx<T> u[16];
#pragma unroll
        for (int b = 0; b < 16 / R3; ++b)
#pragma unroll
            for (int r = 0; r < R3; ++r) {
                const int rp = b + r * (16 / R3);
                const cpx<T> hv = bspec[t + NT * rp];
                u[rp] = cmul(v[b * R3 + r], cpx<T>{hv.x * inv_m, hv.y * inv_m});
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = u[r];
        fft3(std::integral_constant<int, 1>{});
        if (active) {
            cpx<T>* yv = y + vec * (size_t)n;
#pragma unroll
            for (int b = 0; b < 16 / R3; ++b)
#pragma unroll
                for (int r = 0; r < R3; ++r) {
                    const unsigned k = (unsigned)F::template out_index<R3, NSL>(t, b, r);
                    if (k < n) {
                        const cpx<T> z = v[b * R3 + r], cc = c[k];
                        yv[k] = cpx<T>{z.x * cc.x + z.y * cc.y, z.y * cc.x - z.x * cc.y};
                    }
                }
        }
    }
}

template <typename T, int M>
static int launch_bs_wg(const T* x, T* y, const T* c, const T* bspec, size_t n, size_t batch, hipStream_t s)
{
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(M, &wtab));
    constexpr int B = 256 / (M / 16);
    using F = WgFft<T, M, M / 16>;
    const size_t lds = ((size_t)B * (F::LDS_ELEMS + 1) + 16 * 17) * sizeof(cpx<T>);
    auto k = k_bluestein_wg<T, M>;
    if (lds > 64 * 1024)
        BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    size_t groups = (batch + B - 1) / B, slots = (size_t)num_cus() * 4;
    hipLaunchKernelGGL(k, dim3((unsigned)(groups < slots ? groups : slots)), dim3(256), lds, s, reinterpret_cast<const cpx<T>*>(x),
                       reinterpret_cast<cpx<T>*>(y), reinterpret_cast<const cpx<T>*>(c),
                       reinterpret_cast<const cpx<T>*>(bspec), wtab, (unsigned)n, batch);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// x -> y (may alias) for m in {512, 1024, 2048, 4096}; anything else: BDSP_ERR_UNSUPPORTED
template <typename T>
int bs_fused(const T* x, T* y, const T* c, const T* bspec, size_t n, size_t m, size_t batch, hipStream_t s)
{
    switch (m) {
    case 512: return launch_bs_wg<T, 512>(x, y, c, bspec, n, batch, s);
    case 1024: return launch_bs_wg<T, 1024>(x, y, c, bspec, n, batch, s);
    case 2048: return launch_bs_wg<T, 2048>(x, y, c, bspec, n, batch, s);
    case 4096: return launch_bs_wg<T, 4096>(x, y, c, bspec, n, batch, s);
    default: return BDSP_ERR_UNSUPPORTED;
    }
}

static inline unsigned bs_grid(size_t items)
{
    size_t g = (items + 255) / 256, cap = (size_t)num_cus() * 16;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

template <typename T> int bs_chirp(T* c, size_t n, bool inverse, hipStream_t s)
{
    hipLaunchKernelGGL((k_bs_chirp<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<cpx<T>*>(c), (unsigned long long)n, inverse ? 1 : 0);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T> int bs_kernel(const T* c, T* b, size_t n, size_t m, hipStream_t s)
{
    hipLaunchKernelGGL((k_bs_kernel<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const cpx<T>*>(c), reinterpret_cast<cpx<T>*>(b), (unsigned long long)n,
                       (unsigned long long)m);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T>
int bs_pre(const T* x, T* a, const T* c, size_t n, size_t m, size_t batch, bool in_real, T in_scale, size_t rot,
           int window_id, T alpha, hipStream_t s)
{
    dim3 grid(bs_grid(m), (unsigned)batch);
    if (in_real)
        hipLaunchKernelGGL((k_bs_pre<T, true>), grid, dim3(256), 0, s, x, reinterpret_cast<cpx<T>*>(a),
                           reinterpret_cast<const cpx<T>*>(c), (unsigned long long)n, (unsigned long long)m, in_scale,
                           (unsigned long long)rot, window_id, alpha);
    else
        hipLaunchKernelGGL((k_bs_pre<T, false>), grid, dim3(256), 0, s, x, reinterpret_cast<cpx<T>*>(a),
                           reinterpret_cast<const cpx<T>*>(c), (unsigned long long)n, (unsigned long long)m, in_scale,
                           (unsigned long long)rot, window_id, alpha);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
template <typename T>
int bs_post(const T* conv, T* out, const T* c, size_t n, size_t m, size_t batch, int out_kind, size_t rot,
            int div_window_id, T alpha, hipStream_t s)
{
    dim3 grid(bs_grid(n), (unsigned)batch);
#define BDSP_POST(K)                                                                                          \
    hipLaunchKernelGGL((k_bs_post<T, K>), grid, dim3(256), 0, s, reinterpret_cast<const cpx<T>*>(conv), out,  \
                       reinterpret_cast<const cpx<T>*>(c), (unsigned long long)n, (unsigned long long)m,     \
                       (unsigned long long)rot, div_window_id, alpha)
    if (out_kind == 0) BDSP_POST(0);
    else if (out_kind == 1) BDSP_POST(1);
    else BDSP_POST(2);
#undef BDSP_POST
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

#define BDSP_INST(T)                                                                                          \
    template int bs_chirp<T>(T*, size_t, bool, hipStream_t);                                                  \
    template int bs_fused<T>(const T*, T*, const T*, const T*, size_t, size_t, size_t, hipStream_t);          \
    template int bs_kernel<T>(const T*, T*, size_t, size_t, hipStream_t);                                     \
    template int bs_pre<T>(const T*, T*, const T*, size_t, size_t, size_t, bool, T, size_t, int, T, hipStream_t); \
    template int bs_post<T>(const T*, T*, const T*, size_t, size_t, size_t, int, size_t, int, T, hipStream_t);
BDSP_INST(float)
BDSP_INST(double)
#undef BDSP_INST

} // namespace bdsp
