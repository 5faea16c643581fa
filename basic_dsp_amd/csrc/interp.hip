// interp.hip -- InterpolationOps::interpolatef (vector/src/vector_types/time_freq/interpolation.rs:387-482).
//
// Polyphase resampling with wrap-around.  The reference has two code paths with slightly different
// tap windows and this backend follows the same dispatch so the
// results agree with whichever path the CPU would have taken:
//   scalar path  (interpolate_priv_scalar :92-131): any factor; taps evaluated per output
//       y[i] = sum_{m=0}^{2L} x[(r - L + m) mod N] * f(-L - (t - r) + d + m),  t = i/factor, r = floor(t)
//   "simd" path  (interpolate_priv_simd :191-290): integer factor, L <= 202, new_len >= 2000;
//       per-phase tap vectors taps_s[m] = f(-(L-1) + d + m - s/factor) (function_to_vectors :133-181);
//       edges (first/last (2L+1)*factor outputs, interpolate_priv_simd_step :293-315):
//           y[i] = sum_m x[(r - L + 1 + m) mod N] * taps_{i mod f}[m],          r = i div f
//       inner region (register dot product :249-273):
//           y[i] = sum_m x[c + L - 1 - m] * taps_{(f - i mod f) mod f}[m],      c = ceil(i/f)
// All tap arguments are accumulated in T by repeated +1 exactly as the reference does.
#include "bdsp_internal.h"
#include "dsp_funcs.h"

namespace bdsp {

template <typename T> __device__ __forceinline__ T dev_floor(T x);
template <> __device__ __forceinline__ float dev_floor<float>(float x) { return floorf(x); }
template <> __device__ __forceinline__ double dev_floor<double>(double x) { return floor(x); }

template <typename T>
size_t interpolatef_new_len(size_t len, T factor)
{
    // interpolation.rs:406-410: round(len * factor) in T, made even
    T v = (T)len * factor;
    size_t new_len = (size_t)(sizeof(T) == 4 ? roundf((float)v) : round((double)v));
    return new_len + new_len % 2;
}
template size_t interpolatef_new_len<float>(size_t, float);
template size_t interpolatef_new_len<double>(size_t, double);

// taps[s*ntaps + m] = f(-(L-1) + delay + m - s/factor)
template <typename T>
__global__ void k_interp_taps(T* __restrict__ taps, int fid, T rolloff, int conv_len, int factor, T delay)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= factor) return;
    int ntaps = 2 * conv_len + 1;
    T offset = (T)s / (T)factor;
    T j = -((T)conv_len - (T)1) + delay;
    for (int m = 0; m < ntaps; ++m) {
        taps[s * ntaps + m] = conv_time_value<T>(fid, rolloff, j - offset);
        j = j + (T)1;
    }
}

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_interp_table(const T* __restrict__ x, T* __restrict__ y,
                                                       const T* __restrict__ taps, long long points,
                                                       long long new_points, int conv_len, int factor,
                                                       long long skip_lo, long long skip_hi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lt = reinterpret_cast<T*>(smem_raw);
    const int ntaps = 2 * conv_len + 1;
    for (int k = threadIdx.x; k < ntaps * factor; k += blockDim.x) lt[k] = taps[k];
    __syncthreads();
    const long long scalar_len = (long long)ntaps * factor;
    // outputs in [skip_lo, skip_hi) belong to the blocked inner kernel: walk only the two edge runs
    const long long nskip = skip_hi > skip_lo ? skip_hi - skip_lo : 0;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < new_points - nskip;
         g += (long long)gridDim.x * blockDim.x) {
        const long long i = (nskip && g >= skip_lo) ? g + nskip : g;
        T sr = 0, si = 0;
        const bool edge = i < scalar_len || i + scalar_len >= new_points || new_points < 2 * scalar_len;
        if (edge) {
            long long r = i / factor;
            const T* t = lt + (i % factor) * ntaps;
            long long pos = (r - conv_len) % points;
            if (pos < 0) pos += points;
            for (int m = 0; m < ntaps; ++m) {
                pos = pos + 1 < points ? pos + 1 : 0;
                if (CPLX) {
                    T re = x[2 * pos], im = x[2 * pos + 1];
                    sr = sr + (re * t[m] - im * (T)0);
                    si = si + (re * (T)0 + im * t[m]);
                } else sr = sr + x[pos] * t[m];
            }
        } else {
            long long end = (i + factor - 1) / factor + conv_len;
            int shift = (int)((factor - i % factor) % factor);
            const T* t = lt + shift * ntaps;
            for (int m = ntaps - 1; m >= 0; --m) {
                long long n = end - 1 - m;
                if (CPLX) {
                    sr = sr + x[2 * n] * t[m];
                    si = si + x[2 * n + 1] * t[m];
                } else sr = sr + x[n] * t[m];
            }
        }
        if (CPLX) { y[2 * i] = sr; y[2 * i + 1] = si; }
        else y[i] = sr;
    }
}

// Inner region of the integer-factor ("simd") path, register/LDS blocked: a thread owns one input
// position q and produces the FACTOR outputs i = FACTOR*q + s.  With c = ceil(i/f) the reference sums
//   s = 0 :  sum_m x[q + L - 1 - m] * taps_0[m]          s > 0 :  sum_m x[q + L - m] * taps_{f-s}[m]
// (interpolation.rs:249-273), lowest address first.  Walking n = q-L-1 .. q+L once, x[n] feeds output
// s = 0 with tap m = q+L-1-n and outputs s > 0 with tap m = q+L-n, so each input sample is read from
// LDS once for FACTOR outputs (the one-output-per-thread kernel re-reads it FACTOR times through L1
// and measured 14 % of the HBM roofline on config C4).  Results cross threads through LDS so the
// stores are contiguous.  The workgroup covers q in [q0, q0+256) of the inner region
// [q_lo, q_hi) = positions whose FACTOR outputs are all inner outputs.
template <typename T, bool CPLX, int FACTOR>
__global__ __launch_bounds__(256) void k_interp_inner(const T* __restrict__ x, T* __restrict__ y,
                                                      const T* __restrict__ taps, long long q_lo,
                                                      long long q_hi, int conv_len, long long points)
{
    constexpr int E = CPLX ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ntaps = 2 * conv_len + 1;
    T* lt = reinterpret_cast<T*>(smem_raw);                 // [FACTOR][ntaps]
    T* lx = lt + ((FACTOR * ntaps + 1) & ~1);               // [256 + 2L + 2][E]
    T* lo = lx + (256 + 2 * conv_len + 2) * E;              // [256 * FACTOR][E] output staging
    const int t = threadIdx.x;
    const long long q0 = q_lo + (long long)blockIdx.x * 256;
    for (int k = t; k < ntaps * FACTOR; k += 256) lt[k] = taps[k];
    // x[q0 - L - 1 .. q0 + 255 + L]  (always in range: the inner region starts (2L+1) positions in)
    const int span = 256 + 2 * conv_len + 2;
    const long long xbase = q0 - conv_len - 1;
    for (int k = t; k < span * E; k += 256) {
        long long g = xbase * E + k;
        lx[k] = g < points * E ? x[g] : (T)0; // the last workgroup's tile may overhang the vector
    }
    __syncthreads();
    T ar[FACTOR], ai[FACTOR];
#pragma unroll
    for (int s = 0; s < FACTOR; ++s) { ar[s] = 0; ai[s] = 0; }
    // local index of x[n] in lx: n - xbase = (q - q0) + (n - q) + L + 1 = t + j, j = 0 .. 2L+1
    for (int j = 0; j <= 2 * conv_len + 1; ++j) {
        T xr = lx[(t + j) * E], xi = CPLX ? lx[(t + j) * E + 1] : (T)0;
        // n = q - L - 1 + j.  s = 0: m = q+L-1-n = 2L - j (valid for j <= 2L)
        if (j <= 2 * conv_len) {
            T w = lt[2 * conv_len - j];
            ar[0] = ar[0] + xr * w;
            if (CPLX) ai[0] = ai[0] + xi * w;
        }
        // s > 0: m = q+L-n = 2L + 1 - j (valid for j >= 1), tap vector f - s
        if (j >= 1) {
#pragma unroll
            for (int s = 1; s < FACTOR; ++s) {
                T w = lt[(FACTOR - s) * ntaps + 2 * conv_len + 1 - j];
                ar[s] = ar[s] + xr * w;
                if (CPLX) ai[s] = ai[s] + xi * w;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < FACTOR; ++s) {
        lo[(t * FACTOR + s) * E] = ar[s];
        if (CPLX) lo[(t * FACTOR + s) * E + 1] = ai[s];
    }
    __syncthreads();
    long long nq = q_hi - q0;
    if (nq > 256) nq = 256;
    const long long out0 = q0 * FACTOR * E, nout = nq * FACTOR * E;
    for (long long k = t; k < nout; k += 256) y[out0 + k] = lo[k];
}

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_interp_scalar(const T* __restrict__ x, T* __restrict__ y,
                                                        long long points, long long new_points,
                                                        int conv_len, T factor, T delay, int fid, T rolloff)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < new_points;
         i += (long long)gridDim.x * blockDim.x) {
        T center = (T)i / factor;
        T rounded = dev_floor<T>(center);
        long long pos = ((long long)rounded - conv_len - 1) % points;
        if (pos < 0) pos += points;
        T j = -(T)conv_len - (center - rounded) + delay;
        T sr = 0, si = 0;
        for (int k = 0; k < 2 * conv_len + 1; ++k) {
            pos = pos + 1 < points ? pos + 1 : 0;
            T w = conv_time_value<T>(fid, rolloff, j);
            if (CPLX) {
                T re = x[2 * pos], im = x[2 * pos + 1];
                sr = sr + (re * w - im * (T)0);
                si = si + (re * (T)0 + im * w);
            } else sr = sr + x[pos] * w;
            j = j + (T)1;
        }
        if (CPLX) { y[2 * i] = sr; y[2 * i + 1] = si; }
        else y[i] = sr;
    }
}

template <typename T>
int interpolatef_dev(const T* in, T* out, size_t len, bool is_complex, int fid, T rolloff, T factor,
                     T delay, size_t conv_len, T delta, hipStream_t s)
{
    const size_t elem = is_complex ? 2 : 1;
    const size_t points = len / elem;
    if (points == 0) return BDSP_OK;
    delay = delay / delta;                                    // interpolation.rs:397
    if (conv_len > points / 2) conv_len = points / 2;         // :399-404
    const size_t new_len = interpolatef_new_len<T>(len, factor);
    const size_t new_points = new_len / elem;
    if (new_points == 0) return BDSP_OK;
    T rf = sizeof(T) == 4 ? (T)roundf((float)factor) : (T)round((double)factor);
    T dif = rf - factor;
    if (dif < 0) dif = -dif;
    const bool simd = conv_len <= 202 && new_len >= 2000 && dif < (T)1e-6; // :411-414
    size_t blocks = (new_points + 255) / 256;
    size_t cap = (size_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (simd) {
        int f = (int)rf;
        int ntaps = 2 * (int)conv_len + 1;
        WsBlock tb;
        BDSP_TRY(tb.alloc(sizeof(T) * (size_t)ntaps * f, s));
        hipLaunchKernelGGL((k_interp_taps<T>), dim3((f + 63) / 64), dim3(64), 0, s, tb.as<T>(), fid,
                           rolloff, (int)conv_len, f, delay);
        BDSP_LAUNCH_CHECK();
        size_t lds = sizeof(T) * (size_t)ntaps * f;
        if (lds > 60 * 1024) { set_last_error("interpolatef: tap table exceeds LDS"); return BDSP_ERR_UNSUPPORTED; }
        // edges (and everything, for factors without a blocked instantiation) by the generic kernel;
        // the inner region is then overwritten by the blocked kernel where one exists
        const long long scalar_len = (long long)ntaps * f;
        long long q_lo = (scalar_len + f - 1) / f;                   // first q with f*q >= scalar_len
        long long q_hi = ((long long)new_points - scalar_len) / f;   // f*q + f - 1 < new_points - scalar_len
        const bool blocked = (f == 2 || f == 3 || f == 4 || f == 8) && q_hi > q_lo &&
                             (long long)new_points >= 2 * scalar_len;
        long long edge_points = blocked ? 0 : (long long)new_points;
        if (is_complex)
            hipLaunchKernelGGL((k_interp_table<T, true>), dim3((unsigned)blocks), dim3(256), lds, s, in, out,
                               tb.as<T>(), (long long)points, (long long)new_points, (int)conv_len, f,
                               blocked ? q_lo * f : -1LL, blocked ? q_hi * f : -1LL);
        else
            hipLaunchKernelGGL((k_interp_table<T, false>), dim3((unsigned)blocks), dim3(256), lds, s, in, out,
                               tb.as<T>(), (long long)points, (long long)new_points, (int)conv_len, f,
                               blocked ? q_lo * f : -1LL, blocked ? q_hi * f : -1LL);
        (void)edge_points;
        if (blocked) {
            BDSP_LAUNCH_CHECK();
            const int e = is_complex ? 2 : 1;
            size_t lds2 = sizeof(T) * (((size_t)f * ntaps + 1) / 2 * 2 + (256 + 2 * conv_len + 2) * e + 256 * (size_t)f * e);
            unsigned g = (unsigned)((q_hi - q_lo + 255) / 256);
#define BDSP_INNER(FV)                                                                             \
    do {                                                                                           \
        if (is_complex)                                                                            \
            hipLaunchKernelGGL((k_interp_inner<T, true, FV>), dim3(g), dim3(256), lds2, s, in, out, \
                               tb.as<T>(), q_lo, q_hi, (int)conv_len, (long long)points);          \
        else                                                                                       \
            hipLaunchKernelGGL((k_interp_inner<T, false, FV>), dim3(g), dim3(256), lds2, s, in, out, \
                               tb.as<T>(), q_lo, q_hi, (int)conv_len, (long long)points);          \
    } while (0)
            if (f == 2) BDSP_INNER(2); else if (f == 3) BDSP_INNER(3); else if (f == 4) BDSP_INNER(4); else BDSP_INNER(8);
#undef BDSP_INNER
        }
    } else {
        if (is_complex)
            hipLaunchKernelGGL((k_interp_scalar<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, in, out,
                               (long long)points, (long long)new_points, (int)conv_len, factor, delay, fid, rolloff);
        else
            hipLaunchKernelGGL((k_interp_scalar<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, in, out,
                               (long long)points, (long long)new_points, (int)conv_len, factor, delay, fid, rolloff);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template int interpolatef_dev<float>(const float*, float*, size_t, bool, int, float, float, float, size_t, float, hipStream_t);
template int interpolatef_dev<double>(const double*, double*, size_t, bool, int, double, double, double, size_t, double, hipStream_t);

} // namespace bdsp
