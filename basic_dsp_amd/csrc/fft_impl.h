// fft_impl.h -- (included by fft_f32.hip / fft_f64.hip with BDSP_FFT_T set) unnormalised complex FFT kernels for gfx950 (f32 and f64).
//
// Replaces, for the hot path: rustfft behind fft() (vector/src/vector_types/time_freq/mod.rs:32-63)
// and clFFT behind GpuSupport::fft (vector/src/gpu_support/ocl/mod.rs:301-357).
//
// Power-of-two lengths:
//   n <= 8            one thread per transform (k_fft_tiny)
//   16 <= n <= 4096   one workgroup-resident Stockham FFT, 16 points per thread, data crosses
//                     threads through LDS only (k_fft_wg); several small transforms per workgroup
//   n > 4096          2 or 3 global Stockham passes of super-radix RP in {64..1024} (k_fft_pass):
//                     each workgroup takes a tile of W = 4096/RP adjacent columns, so every global
//                     access is a run of W contiguous points (RP = 256: 128-byte segments), does the
//                     RP-point sub-FFT for all W columns in registers + LDS, applies the inter-pass
//                     twiddle on load and writes the autosorted result.  2^24 points = 3 passes.
// Every other length goes through Bluestein's chirp-z on the same kernels (fft_any).
// Window, 1/N scale, fft_shift / ifft_shift and magnitude are fused into the first / last pass
// (FftIo), so fft()/windowed_fft()/ifft() never take an extra trip through HBM
// (reference: time_to_freq.rs:158-175, freq_to_time.rs:160-177 run them as separate passes).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "bdsp_internal.h"
#include "dsp_funcs.h"

// BDSP_FFT_PART (round 6).  A process's first transform loads the code object of the translation unit its first kernel lives
// in (HIP loads per unit: *measured*, tools/first_call_probe.py -- 4.2 ms for the 2.7 MB of all f32 kernels), and two thirds
// of that code serves fused options and generic I/O that a plain transform never launches.  So each precision is TWO units:
//   1  "plain" (fft_f32.hip / fft_f64.hip): plan selection, fft_pow2, and every kernel a plain transform can launch -- the
//      SIMPLE pass instantiations, k_fft_wg without generic I/O, k_fft_wg_batch, k_fft_wg4, k_fft_tiny;
//   2  "options" (fft_f32_opts.hip / fft_f64_opts.hip): the pass kernels with fused shift / scale / window / real input /
//      magnitude / real-part output and the generic-I/O (GEN) kernels, reached through launch_pass_opts_rp / launch_wg_gen.
//   0  everything in one unit.
#ifndef BDSP_FFT_PART
#define BDSP_FFT_PART 0
#endif

namespace bdsp {

template <typename T>
int launch_pass_opts_rp(int rp, int w, const FftIo<T>& io, const cpx<T>* src, cpx<T>* dst, size_t n, size_t nsg, size_t batch,
                        bool inverse, bool first, bool last, hipStream_t s, int tl, int aux);
template <typename T>
int launch_wg_gen(int n, const FftIo<T>& io, size_t batch, bool inverse, hipStream_t s);

// ------------------------------------------------------------------------------ fused I/O
template <typename T>
__device__ __forceinline__ T dev_hypot(T a, T b);
template <> __device__ __forceinline__ float dev_hypot<float>(float a, float b) { return hypotf(a, b); }
template <> __device__ __forceinline__ double dev_hypot<double>(double a, double b) { return hypot(a, b); }

template <typename T>
__device__ __forceinline__ cpx<T> io_load(const FftIo<T>& io, size_t vec, size_t i)
{
    size_t src = i;
    if (io.flags & BDSP_FFT_SHIFT_IN) { // ifft_shift: out[i] = in[(i + floor(n/2)) mod n]
        src = i + io.n / 2;
        if (src >= io.n) src -= io.n;
    }
    cpx<T> v;
    if (io.in_valid && src >= io.in_valid) return cpx<T>{0, 0};
    if (io.flags & FFT_IN_REAL) {
        v.x = reinterpret_cast<const T*>(io.in)[vec * io.in_stride + src];
        v.y = (T)0;
    } else {
        v = reinterpret_cast<const cpx<T>*>(io.in)[vec * io.in_stride + src];
    }
    if (io.window_id >= 0 && !(io.flags & FFT_WINDOW_OUT_DIV)) {
        T w = window_value_sym<T>(io.window_id, io.window_alpha, src, io.n);
        v.x = v.x * w;
        v.y = v.y * w;
    }
    if (io.in_scale != (T)1) {
        v.x = v.x * io.in_scale;
        v.y = v.y * io.in_scale;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ void io_store(const FftIo<T>& io, size_t vec, size_t k, cpx<T> v)
{
    size_t dst = k;
    if (io.flags & BDSP_FFT_SHIFT_OUT) { // fft_shift: out[i] = in[(i + ceil(n/2)) mod n]
        dst = k + io.n / 2;
        if (dst >= io.n) dst -= io.n;
    }
    if (io.flags & FFT_WINDOW_OUT_DIV) {
        T w = (T)1 / window_value_sym<T>(io.window_id, io.window_alpha, dst, io.n);
        v.x = v.x * w;
        v.y = v.y * w;
    }
    if (io.flags & BDSP_FFT_MAGNITUDE)
        reinterpret_cast<T*>(io.out)[vec * io.out_stride + dst] = dev_hypot<T>(v.x, v.y);
    else if (io.flags & FFT_OUT_REAL)
        reinterpret_cast<T*>(io.out)[vec * io.out_stride + dst] = v.x;
    else
        reinterpret_cast<cpx<T>*>(io.out)[vec * io.out_stride + dst] = v;
}

// LDS stride (in elements) between the regions of adjacent columns / transforms of one workgroup.
// Lanes that run along columns hit addresses c*stride + const: with the padded length alone the
// stride is a multiple of 16 elements = a multiple of 32 banks, i.e. a 16-way conflict (measured:
// 90 % of LDS cycles were conflict cycles).  stride = 16*k + max(1, 16/W) spreads the W columns of a
// 16-lane group over all banks, and leaves room for the rows of a second thread-row when W < 16.
__host__ __device__ constexpr int col_stride(int n, int w = 0)
{
    int nt = n / 16 > 0 ? n / 16 : 1;
    if (w == 0) w = 256 / nt;
    int padded = n + (n >> 4);
    int base = (padded + 15) / 16 * 16;
    return base + (w >= 16 ? 1 : 16 / w);
}

// Inner-stage twiddles of a pass kernel from an LDS copy of the RP-entry table instead of global memory?  Yes whenever
// the copy does not cost a resident workgroup.  *Measured* on 64 x 2^20 points (two passes of 1024-point columns):
// 479 us with the 27 dependent global loads per thread and tile between an LDS gather and its butterflies, 405 us with
// the loads removed altogether (a timing experiment); keeping the twiddles in registers across a persistent tile
// loop instead made hipcc allocate 200+ VGPRs (532 us) or, with the budget capped at 128, spill (890 us).
__host__ __device__ constexpr int pass_wgs_per_cu(size_t lds_bytes, int threads)
{
    size_t k = (size_t)(160 * 1024) / (lds_bytes + 64);
    if (k > (size_t)(2048 / threads)) k = (size_t)(2048 / threads);
    if (k > 8) k = 8;
    return (int)k;
}
// SPLIT exchange (round 3): an f64 tile of 2048 x 4 or 1024 x 8 points is 136 KB of LDS -- ONE 512-thread workgroup per
// CU, which loads, transforms and stores its tile with nothing to overlap any phase (config C4a's 4M points are two such
// tiles per CU in all).  These tiles therefore cross their real and imaginary parts one after the other through a
// buffer of HALF the size (8-byte elements: twice the LDS instructions and barriers, the same bytes), which lets two
// workgroups share a CU.  Not for the GEN instantiations, which stage whole complex values through the buffer.
template <typename T, int RP, int W, bool GEN>
constexpr bool pass_split_exchange()
{
#if defined(BDSP_LAB) && defined(BDSP_FFT_NO_SPLIT)
    return false; // (A/B build: tools/plan_matrix.sh, round 5 -- the whole-complex exchange of round 2, one workgroup per CU)
#endif
    return !GEN && sizeof(T) == 8 && (size_t)W * col_stride(RP, W) * sizeof(cpx<T>) > 80 * 1024;
}
template <typename T, int RP, int W, bool GEN>
constexpr size_t pass_tile_lds_bytes()
{
    return (size_t)W * col_stride(RP, W) * (pass_split_exchange<T, RP, W, GEN>() ? sizeof(T) : sizeof(cpx<T>));
}

// waves per SIMD the pass kernel is compiled for: the split tiles want two 512-thread workgroups per CU = 4 (128 VGPRs)
template <typename T, int RP, int W, bool GEN>
constexpr int pass_min_waves()
{
    return pass_split_exchange<T, RP, W, GEN>() ? 2 * (W * (RP / 16)) / 256 : 1;
}

template <typename T, int RP, int W, bool GEN = true>
constexpr bool pass_lds_twiddles()
{
    constexpr size_t base = pass_tile_lds_bytes<T, RP, W, GEN>(), tab = (size_t)RP * sizeof(cpx<T>);
    // ... and always for the 4-wide 1024-point f32 tiles: the plan picks them only when the launch has fewer than two
    // tiles per CU (one 2^20-point vector: 256 tiles), where the fourth resident workgroup the table costs is never there
    if (sizeof(T) == 4 && RP == 1024 && W == 4) return true;
    return RP >= 256 && pass_wgs_per_cu(base, W * (RP / 16)) == pass_wgs_per_cu(base + tab, W * (RP / 16));
}

// ------------------------------------------------------------------------------ n <= 8
template <typename T, int N, int DIR>
__global__ __launch_bounds__(256) void k_fft_tiny(FftIo<T> io, size_t batch)
{
    size_t vec = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec >= batch) return;
    cpx<T> v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = io_load(io, vec, i);
    dft<N, DIR>(v);
#pragma unroll
    for (int i = 0; i < N; ++i) io_store(io, vec, i, v[i]);
}

// ------------------------------------------------------------------------------ 16 <= n <= 4096
// GEN = false: plain complex in/out, registers <-> global directly.
// GEN = true : any fused option (window, scale, shifts, real input, magnitude ...).  The option
// code runs in a ROLLED loop that stages through LDS, so the register array stays statically
// indexed and the option code is emitted once instead of 16 times per thread.
template <typename T, int N, int DIR, bool GEN>
__global__ __launch_bounds__(256) void k_fft_wg(FftIo<T> io, const cpx<T>* __restrict__ wtab,
                                                 size_t batch)
{
    constexpr int NT = N / 16;
    constexpr int B = 256 / NT; // transforms per workgroup
    using F = WgFft<T, N, NT>;
    using P = Radix16Plan<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem_raw);

    const int tid = threadIdx.x;
    const int c = tid / NT, t = tid % NT;
    const size_t vec = (size_t)blockIdx.x * B + c;
    const bool active = vec < batch;
    cpx<T>* l = lds + (size_t)c * col_stride(N);
    auto tw = [&](int m) { return wtab[m]; };

    cpx<T> v[16];
    if constexpr (GEN) {
#pragma unroll 2
        for (int e = 0; e < 16; ++e) {
            int idx = t + e * NT;
            l[F::pad(idx)] = active ? io_load(io, vec, (size_t)idx) : cpx<T>{0, 0};
        }
        __syncthreads();
        if constexpr (N >= 256) F::template gather<16>(v, t, l);
        else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = l[F::pad(F::template in_index<16>(t, 0, r))];
        }
        __syncthreads();
    } else {
        const cpx<T>* in = reinterpret_cast<const cpx<T>*>(io.in) + vec * io.in_stride;
        const int nvalid = io.in_valid ? (int)io.in_valid : N;
        // ifft_shift of the input (even n): in[(i + n/2) mod n] with i = t + r*NT is register r ^ 8's address
        const int rx = (io.flags & BDSP_FFT_SHIFT_IN) ? 8 : 0;
        if (io.flags & FFT_IN_REAL) {
            const T* inr = reinterpret_cast<const T*>(io.in) + vec * io.in_stride;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = F::template in_index<16>(t, 0, r ^ rx);
                v[r] = cpx<T>{(active && idx < nvalid) ? inr[idx] : (T)0, (T)0};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = F::template in_index<16>(t, 0, r ^ rx);
                v[r] = (active && idx < nvalid) ? in[idx] : cpx<T>{0, 0};
            }
        }
        if (io.in_scale != (T)1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cpx<T>{v[r].x * io.in_scale, v[r].y * io.in_scale};
        }
    }
    F::template compute<16, 1, DIR>(v, t, tw);
    if constexpr (P::R2 > 1) {
        F::template scatter<16, 1>(v, t, l);
        __syncthreads();
        F::template gather<P::R2>(v, t, l);
        F::template compute<P::R2, 16, DIR>(v, t, tw);
    }
    if constexpr (P::R3 > 1) {
        __syncthreads();
        F::template scatter<P::R2, 16>(v, t, l);
        __syncthreads();
        F::template gather<P::R3>(v, t, l);
        F::template compute<P::R3, 16 * P::R2, DIR>(v, t, tw);
    }
    constexpr int RL = P::R3 > 1 ? P::R3 : (P::R2 > 1 ? P::R2 : 16);
    constexpr int NSL = N / RL;
    if constexpr (GEN) {
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16 / RL; ++b)
#pragma unroll
            for (int r = 0; r < RL; ++r) l[F::pad(F::template out_index<RL, NSL>(t, b, r))] = v[b * RL + r];
        __syncthreads();
        if (!active) return;
#pragma unroll 1
        for (int e = 0; e < 16; ++e) {
            int idx = t + e * NT;
            io_store(io, vec, (size_t)idx, l[F::pad(idx)]);
        }
    } else {
        if (!active) return;
        cpx<T>* out = reinterpret_cast<cpx<T>*>(io.out) + vec * io.out_stride;
        // fft_shift of the output (even n): the last stage's digit r is the top digit of the output index
        const int sx = (io.flags & BDSP_FFT_SHIFT_OUT) ? RL / 2 : 0;
        if (io.flags & (BDSP_FFT_MAGNITUDE | FFT_OUT_REAL)) {
            T* outr = reinterpret_cast<T*>(io.out) + vec * io.out_stride;
            const bool mag = (io.flags & BDSP_FFT_MAGNITUDE) != 0;
#pragma unroll
            for (int b = 0; b < 16 / RL; ++b)
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const cpx<T> z = v[b * RL + r];
                    outr[F::template out_index<RL, NSL>(t, b, r ^ sx)] = mag ? dev_hypot<T>(z.x, z.y) : z.x;
                }
            return;
        }
#pragma unroll
        for (int b = 0; b < 16 / RL; ++b)
#pragma unroll
            for (int r = 0; r < RL; ++r) out[F::template out_index<RL, NSL>(t, b, r ^ sx)] = v[b * RL + r];
    }
}

// Persistent batched variant of k_fft_wg for N in {1024, 2048, 4096}, plain I/O, large batches: a
// workgroup walks the batch with a grid stride, keeps the last stage's twiddles in registers and the
// 16 x 15 second-stage twiddles in an LDS table (k_fft_wg re-reads both from L2 for every transform,
// twice in its dependency chain), and uses the conflict-free exchange layouts for N = 4096.
// *Measured* (16M points in all): 4096 x 4096: 75.9 -> 53.2 us, 2048 x 8192: 66.9 -> 54.4 us,
// 1024 x 16384: 62.7 -> 54.0 us (5.0 TB/s algorithmic).
template <typename T, int N, int DIR>
__global__ __launch_bounds__(256) void k_fft_wg_batch(FftIo<T> io, const cpx<T>* __restrict__ wtab, size_t batch)
{
    constexpr int NT = N / 16;
    constexpr int B = 256 / NT;
    using F = WgFft<T, N, NT>;
    using P = Radix16Plan<N>;
    static_assert(P::R2 == 16 && P::R3 >= 4, "1024 <= N <= 4096");
    constexpr int R3 = P::R3, NTW3 = (16 / R3) * (R3 - 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem_raw);
    const int tid = threadIdx.x;
    const int c = tid / NT, t = tid % NT;
    cpx<T>* l = lds + (size_t)c * col_stride(N);
    cpx<T>* tw2l = lds + (size_t)B * col_stride(N);
    auto tw = [&](int m) { return wtab[m]; };

    // R3 = 16 (N = 4096): six values instead of fifteen (w^r = w^(4a) w^b) keep the kernel at 4 per CU
    cpx<T> tw3[R3 == 16 ? 1 : NTW3], tw3a[3], tw3b[3];
    if constexpr (R3 == 16) F::template load_twiddles16_split<256>(tw3a, tw3b, t, tw);
    else F::template load_twiddles<R3, 256>(tw3, t, tw);
    if (tid < 240) {
        int k = tid / 15, r = tid % 15 + 1;
        tw2l[k * 17 + r - 1] = wtab[r * k * (N / 256)];
    }
    __syncthreads();
    const cpx<T>* tw2p = tw2l + (t & 15) * 17;
    const int nvalid = io.in_valid ? (int)io.in_valid : N;
    const size_t groups = (batch + B - 1) / B;
    for (size_t gi = blockIdx.x; gi < groups; gi += gridDim.x) {
        const size_t vec = gi * B + c;
        const bool active = vec < batch;
        const cpx<T>* in = reinterpret_cast<const cpx<T>*>(io.in) + vec * io.in_stride;
        cpx<T> v[16];
        const int rx = (io.flags & BDSP_FFT_SHIFT_IN) ? 8 : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = t + (r ^ rx) * NT;
            v[r] = (active && idx < nvalid) ? in[idx] : cpx<T>{0, 0};
        }
        if (io.in_scale != (T)1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cpx<T>{v[r].x * io.in_scale, v[r].y * io.in_scale};
        }
        F::template compute<16, 1, DIR>(v, t, tw);
        __syncthreads(); // the previous transform's last gather is done
        if constexpr (N == 4096) F::scatter_a(v, t, l); else F::template scatter<16, 1>(v, t, l);
        __syncthreads();
        if constexpr (N == 4096) F::gather_a(v, t, l); else F::template gather<16>(v, t, l);
        F::template compute_pre<16, 16, DIR>(v, tw2p);
        __syncthreads();
        if constexpr (N == 4096) F::scatter_b(v, t, l); else F::template scatter<16, 16>(v, t, l);
        __syncthreads();
        if constexpr (N == 4096) F::gather_b(v, t, l); else F::template gather<R3>(v, t, l);
        if constexpr (R3 == 16) F::template compute_pre16_split<256, DIR>(v, tw3a, tw3b);
        else F::template compute_pre<R3, 256, DIR>(v, tw3);
        if (active) {
            cpx<T>* out = reinterpret_cast<cpx<T>*>(io.out) + vec * io.out_stride;
            const int sx = (io.flags & BDSP_FFT_SHIFT_OUT) ? R3 / 2 : 0;
#pragma unroll
            for (int b = 0; b < 16 / R3; ++b)
#pragma unroll
                for (int r = 0; r < R3; ++r) out[F::template out_index<R3, N / R3>(t, b, r ^ sx)] = v[b * R3 + r];
        }
    }
}

// 8192-point transforms in ONE workgroup (f32, plain I/O): 512 threads x 16 points,
// four stages 16 x 16 x 16 x 2 (the template also covers 16384 = ... x 4 with 1024 threads), three LDS exchanges -- one trip through HBM (16 B per point) where the
// two-pass plan makes two, and one launch instead of two for a single transform.  Persistent over the batch;
// second-stage twiddles in an LDS table, last-stage twiddles in registers, third-stage twiddles from L2.
template <typename T, int N, int DIR>
__global__ __launch_bounds__(N / 16) void k_fft_wg4(FftIo<T> io, const cpx<T>* __restrict__ wtab, size_t batch)
{
    constexpr int NT = N / 16, R4 = N / 4096, NTW4 = (16 / R4) * (R4 - 1);
    static_assert(R4 == 2 || R4 == 4, "N in {8192, 16384}");
    using F = WgFft<T, N, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* l = reinterpret_cast<cpx<T>*>(smem_raw);
    cpx<T>* tw2l = l + F::LDS_ELEMS;
    const int t = threadIdx.x;
    auto tw = [&](int m) { return wtab[m]; };
    cpx<T> tw4[NTW4];
    F::template load_twiddles<R4, 4096>(tw4, t, tw);
    if (t < 240) {
        int k = t / 15, r = t % 15 + 1;
        tw2l[k * 17 + r - 1] = wtab[r * k * (N / 256)];
    }
    __syncthreads();
    const cpx<T>* tw2p = tw2l + (t & 15) * 17;
    const int nvalid = io.in_valid ? (int)io.in_valid : N;
    for (size_t vec = blockIdx.x; vec < batch; vec += gridDim.x) {
        const cpx<T>* in = reinterpret_cast<const cpx<T>*>(io.in) + vec * io.in_stride;
        cpx<T> v[16];
        const int rx = (io.flags & BDSP_FFT_SHIFT_IN) ? 8 : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = t + (r ^ rx) * NT;
            v[r] = idx < nvalid ? in[idx] : cpx<T>{0, 0};
        }
        if (io.in_scale != (T)1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cpx<T>{v[r].x * io.in_scale, v[r].y * io.in_scale};
        }
        F::template compute<16, 1, DIR>(v, t, tw);
        __syncthreads(); // the previous transform's last gather is done
        F::template scatter<16, 1>(v, t, l);
        __syncthreads();
        F::template gather<16>(v, t, l);
        F::template compute_pre<16, 16, DIR>(v, tw2p);
        __syncthreads();
        F::template scatter<16, 16>(v, t, l);
        __syncthreads();
        F::template gather<16>(v, t, l);
        F::template compute<16, 256, DIR>(v, t, tw);
        __syncthreads();
        F::template scatter<16, 256>(v, t, l);
        __syncthreads();
        F::template gather<R4>(v, t, l);
        F::template compute_pre<R4, 4096, DIR>(v, tw4);
        cpx<T>* out = reinterpret_cast<cpx<T>*>(io.out) + vec * io.out_stride;
        const int sx = (io.flags & BDSP_FFT_SHIFT_OUT) ? R4 / 2 : 0;
#pragma unroll
        for (int b = 0; b < 16 / R4; ++b)
#pragma unroll
            for (int r = 0; r < R4; ++r) out[F::template out_index<R4, N / R4>(t, b, r ^ sx)] = v[b * R4 + r];
    }
}

// ------------------------------------------------------------------------------ n > 4096
// exp(-2*pi*i*e/n) for an exact integer e < n (n a power of two): the argument 2e/n is exact in
// float for n <= 2^24 and always exact in double, so the only error is sincospi's own.
template <typename T>
__device__ __forceinline__ cpx<T> unit_root(size_t e, size_t n);
template <>
__device__ __forceinline__ cpx<float> unit_root<float>(size_t e, size_t n)
{
    if (n <= (size_t(1) << 24)) {
        float s, c;
        sincospif((float)e * (2.0f / (float)n), &s, &c);
        return {c, -s};
    }
    double s, c;
    sincospi((double)e * (2.0 / (double)n), &s, &c);
    return {(float)c, (float)-s};
}
template <>
__device__ __forceinline__ cpx<double> unit_root<double>(size_t e, size_t n)
{
    double s, c;
    sincospi((double)e * (2.0 / (double)n), &s, &c);
    return {c, -s};
}

// One global Stockham iteration of super-radix RP on vectors of n points:
//   column j (0 <= j < n/RP), k = j mod nsg:
//     v[row] = in[j + row*n/RP] * w_{nsg*RP}^{row*k}            row = 0..RP-1
//     v      = DFT_RP(v)
//     out[(j/nsg)*nsg*RP + k + row*nsg] = v[row]
// A workgroup (256 threads, 16 points each) owns W = 4096/RP adjacent columns.
// ROWMAP selects how the LAST inner stage maps lanes: along rows (first global pass, where each
// column's RP results are contiguous in memory) or along columns (later passes, where adjacent
// columns are contiguous).
// GEN: fused options on the vector's input (first pass = the ROWMAP instantiation) or output (last
// pass = a !ROWMAP instantiation), staged through LDS by a rolled loop like k_fft_wg<GEN>.
// SIMPLE: plain complex input without scale / window / shift / real input on the first pass and a plain complex output
// on a later one -- the instantiation the headline transforms take.  The run-time option checks around the loads and
// stores are then compiled out: the branches themselves are free, but hipcc's code around their merge points is not
// (the same lesson as in conv_v2.hip).
// The reference's windows on the sixteen values a pass thread holds, in registers (round 4).  In the first pass
// (input side) and in the last one (output side) a thread's sixteen points form the lattice i0 + m n/16, m = 0..15, with
// i0 < n/16 (rows are n/16 points apart); lattice(reg) says which m register `reg` holds.  Evaluated symmetrically like the
// reference (vector_types/mod.rs:567-594: w(i) = w(n-1-i)): point m >= 8 takes the value of its mirror image
// i1 + (15 - m) n/16, i1 = n/16 - 1 - i0.
//   Hamming / Hann (id 1) and Blackman-Harris (id 2): cos(2 pi i/(n-1)) = cos(theta_b + mm D) from TWO small base angles
//   per thread and the eight constants cos / sin(mm D) the launcher put into io (before: sixteen cospi per thread in f32,
//   three sincospi + sixteen chained rotations in f64, and Blackman-Harris through the staged generic loop with three
//   cospi per ELEMENT); Blackman-Harris's harmonics by cos 2t = 2c^2 - 1, cos 3t = c (4c^2 - 3).
//   Triangular (id 0): the reference's formula, 1 - |(jm - (n-1)/2) / (n/2)|, on the mirrored index jm.
// INV: divide by the window instead (windowed_ifft's un-windowing of the output, time.rs:50-66).
// FIXED >= 0: the window id is a compile-time constant.  The f64 tiles with the split exchange live at a 128-register
// budget; with the code of all three windows in one kernel they spilled 24-54 registers, so THEIR kernels are instantiated
// per window (k_fft_pass WIN) and each carries one loop.
template <typename T, int FIXED, bool INV, class LATTICE>
__device__ __forceinline__ void pass_window16(const FftIo<T>& io, size_t i0, size_t n, cpx<T> (&v)[16], LATTICE lattice)
{
    const size_t i1 = n / 16 - 1 - i0;
    auto put = [&](int reg, T w) {
        if (INV) w = (T)1 / w;
        v[reg] = cpx<T>{v[reg].x * w, v[reg].y * w};
    };
    const int wid = FIXED >= 0 ? FIXED : io.window_id;
    if (wid == 0) {
        const T b0 = ((T)i0 - ((T)n - (T)1) / (T)2) / ((T)n / (T)2), b1 = ((T)i1 - ((T)n - (T)1) / (T)2) / ((T)n / (T)2);
        const T step = (T)0.125; // (n/16) / (n/2)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = lattice(reg);
            put(reg, (T)1 - dev_abs((m < 8 ? b0 : b1) + step * (T)(m < 8 ? m : 15 - m)));
        }
        return;
    }
    const T two_over = (T)2 / ((T)n - (T)1);
    T s0, c0, s1, c1;
    dev_sincospi<T>((T)i0 * two_over, &s0, &c0);
    dev_sincospi<T>((T)i1 * two_over, &s1, &c1);
    if (wid == 1) {
        const T beta = (T)1 - io.window_alpha;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = lattice(reg), mm = m < 8 ? m : 15 - m;
            const T c = (m < 8 ? c0 : c1) * io.win_c[mm] - (m < 8 ? s0 : s1) * io.win_s[mm];
            put(reg, io.window_alpha - beta * c);
        }
    } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = lattice(reg), mm = m < 8 ? m : 15 - m;
            const T c = (m < 8 ? c0 : c1) * io.win_c[mm] - (m < 8 ? s0 : s1) * io.win_s[mm];
            const T c2 = c * c;
            put(reg, (T)0.35875 - (T)0.48829 * c + (T)0.14128 * ((T)2 * c2 - (T)1) - (T)0.01168 * (c * ((T)4 * c2 - (T)3)));
        }
    }
}

// TL (round 4): the TILED intermediate of a two-pass plan.  The first pass's output layout is nobody's business but the
// second pass's, which reads, per tile, W2 adjacent columns of it: W2 x 8 (16) bytes per row -- 32-64-byte runs wherever
// the columns are long (1024 / 2048 points).  With mid'[k1 / W2][j][k1 % W2] (k1 = the first pass's output index = the
// second pass's column, j = the first pass's column = the second pass's row) a second-pass tile is ONE contiguous
// RP2 x W2 block, and the first pass still stores whole lines (W1 x W2 adjacent values per k1 group, lanes along the
// columns).  TL = 1: first pass, tiled store (aux = log2 W2); TL = 2: last pass, tiled load.  The natural-order input of
// the first pass and output of the last keep their W-wide runs.
// WIN: -1 = the window id is read at run time (and is Hamming / Hann where the tile uses the split exchange); 0 / 2: the
// split-exchange tiles' instantiations for the triangular and the Blackman-Harris window (pass_window16 FIXED).
// NTL (round 5): the FIRST pass reads its input -- dead once read -- with non-temporal loads.  f64 only, and only where the
// tile's runs are whole 128-byte lines (pass_ntl_candidate): see launch_pass for the measurements.
template <typename T, int RP, int W, int DIR, bool ROWMAP, bool GEN, bool SIMPLE = false, int TL = 0, int WIN = -1, bool NTL = false>
__global__ __launch_bounds__(W * (RP / 16), (pass_min_waves<T, RP, W, GEN>())) void k_fft_pass(FftIo<T> io, const cpx<T>* src, // (src may equal dst: the in-place last pass)
                                                   cpx<T>* dst,
                                                   const cpx<T>* __restrict__ wtab, size_t n,
                                                   size_t nsg, size_t tiles_per_vec, int last, int aux)
{
    static_assert(TL == 0 || (TL == 1 && ROWMAP) || (TL == 2 && !ROWMAP), "tiled store: first pass; tiled load: a later pass");
    constexpr int NT = RP / 16;
    constexpr int CS = col_stride(RP, W);
    using F = WgFft<T, RP, NT>;
    using P = Radix16Plan<RP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool SPLIT = pass_split_exchange<T, RP, W, GEN>();
    using LE = typename std::conditional<SPLIT, T, cpx<T>>::type; // what an LDS slot holds
    LE* lds = reinterpret_cast<LE*>(smem_raw);

    const int tid = threadIdx.x;
    size_t blk = blockIdx.x;
    const size_t vec = blk / tiles_per_vec;
    size_t tile = blk % tiles_per_vec;
    // XCD-aware placement: block b runs on XCD b % 8; give each XCD a contiguous run of tiles so
    // neighbouring tiles (which share cache lines when W*sizeof(cpx) < 128 B) meet in one L2.
    if ((tiles_per_vec & 7) == 0) tile = (tile & 7) * (tiles_per_vec >> 3) + (tile >> 3);
    const size_t j0 = tile * W;
    const size_t stride_in = n / RP;
    constexpr bool LTW = pass_lds_twiddles<T, RP, W, GEN>();
    static_assert(!(LTW && SPLIT), "a split tile has no room for the table");
    cpx<T>* ltw = reinterpret_cast<cpx<T>*>(lds + (size_t)W * CS); // [RP] copy of the twiddle table (visible after the first barrier below)
    if constexpr (LTW) {
        for (int i = tid; i < RP; i += W * NT) ltw[i] = wtab[i];
    }
    auto tw = [&](int m) { return LTW ? ltw[m] : wtab[m]; };

    // ---- load (lanes along columns: W contiguous points per row)
    const int c = tid % W, ti = tid / W;
    const size_t j = j0 + c;
    cpx<T> v[16];
    if constexpr (GEN && ROWMAP) {
#pragma unroll 1
        for (int e = 0; e < 16; ++e) {
            int row = ti + e * NT;
            lds[(size_t)c * CS + F::pad(row)] = io_load(io, vec, j + (size_t)row * stride_in);
        }
        __syncthreads();
        F::template gather<16>(v, ti, lds + (size_t)c * CS);
        __syncthreads();
    } else {
        const cpx<T>* in = src + vec * n + j;
        // first pass (ROWMAP): ifft_shift = the row index's top bit flipped = register r ^ 8's address
        const int rx = (!SIMPLE && ROWMAP && (io.flags & BDSP_FFT_SHIFT_IN)) ? 8 : 0;
        if (!SIMPLE && ROWMAP && (io.flags & FFT_IN_REAL)) {
            // real input: `points` scalars per vector, imaginary parts are zero (time_to_freq.rs:147-150)
            const T* inr = reinterpret_cast<const T*>(io.in) + vec * io.in_stride + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cpx<T>{inr[(size_t)(ti + (r ^ rx) * NT) * stride_in], (T)0};
        } else if constexpr (TL == 2) {
            const cpx<T>* in_t = src + vec * n + j0 * RP + c; // the tile's own RP x W block
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = in_t[(size_t)(ti + r * NT) * W];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                // (non-temporal loads, -DBDSP_FFT_NTLOAD in a LAB build: *measured* round 4, see DESIGN.md 4.2;
                // -DBDSP_FFT_NTLOAD=2: in the FIRST pass only, whose input is dead once read -- round 5)
#if defined(BDSP_LAB) && defined(BDSP_FFT_NTLOAD)
                v[r] = (BDSP_FFT_NTLOAD != 2 || ROWMAP) ? nt_load(&in[(size_t)(ti + (r ^ rx) * NT) * stride_in]) : in[(size_t)(ti + (r ^ rx) * NT) * stride_in];
#else
                v[r] = NTL ? nt_load(&in[(size_t)(ti + (r ^ rx) * NT) * stride_in]) : in[(size_t)(ti + (r ^ rx) * NT) * stride_in];
#endif
        }
        if (!SIMPLE && ROWMAP && io.in_scale != (T)1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cpx<T>{v[r].x * io.in_scale, v[r].y * io.in_scale};
        }
        if (!SIMPLE && ROWMAP && io.window_id >= 0 && io.window_id <= 2 && !(io.flags & FFT_WINDOW_OUT_DIV)) {
            // the window on the input, in registers: register r holds row q = r ^ rx (rx is 0 or 8), i.e. lattice point q
            pass_window16<T, (SPLIT ? (WIN >= 0 ? WIN : 1) : -1), false>(io, j + (size_t)ti * stride_in, n, v, [&](int r) { return r ^ rx; });
        }
    }
    if (!ROWMAP) { // (nsg > 1 exactly on the later passes)
        // inter-pass twiddle w_n^{row*q}, q = k * n/(nsg*RP); row = ti + r*NT: register r carries bs * d1^r with
        // bs = w_n^(ti q), d1 = w_n^(NT q) -- i.e. the first inner stage IS a twiddled 16-point transform with base
        // d1 on inputs scaled by bs (round 3: dft16_tw from d1, d1^2 and their products with constants; before, fifteen
        // powers of d1 were built and multiplied in one by one: 44 complex multiplies and 64 more f64 registers)
        const size_t k = j % nsg;
        const size_t q = k * (n / (nsg * RP));
        const cpx<T> bs = unit_root<T>(((size_t)ti * q) & (n - 1), n);
        const cpx<T> d1 = unit_root<T>(((size_t)NT * q) & (n - 1), n);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = twmul<DIR>(v[r], bs);
        const cpx<T> held[2] = {cmul(d1, d1), d1};
        cpx<T> t8[8];
        expand_twiddles16_fma<2>(held, t8);
        dft16_tw<DIR>(&v[0], t8);
    } else {
        F::template compute<16, 1, DIR>(v, ti, tw);
    }

    // ---- RP-point sub-FFT of every column
    const int c2 = (ROWMAP && TL != 1) ? tid / NT : tid % W; // (a tiled store wants its lanes along the columns)
    const int t2 = (ROWMAP && TL != 1) ? tid % NT : tid / W;
    LE* l1 = lds + (size_t)c * CS;
    LE* l2 = lds + (size_t)c2 * CS;
    if constexpr (SPLIT) {
        // real parts, then imaginary parts, through the half-size buffer (straight from / into the registers' halves)
        F::template scatter<16, 1>(PartRef<cpx<T>, false>{v}, ti, l1);
        __syncthreads();
        F::template gather<P::R2>(PartRef<cpx<T>, false>{v}, t2, l2);
        __syncthreads();
        F::template scatter<16, 1>(PartRef<cpx<T>, true>{v}, ti, l1);
        __syncthreads();
        F::template gather<P::R2>(PartRef<cpx<T>, true>{v}, t2, l2);
    } else {
        F::template scatter<16, 1>(v, ti, l1);
        __syncthreads();
        F::template gather<P::R2>(v, t2, l2);
    }
    F::template compute<P::R2, 16, DIR>(v, t2, tw);
    if constexpr (P::R3 > 1) {
        __syncthreads();
        if constexpr (SPLIT) {
            F::template scatter<P::R2, 16>(PartRef<cpx<T>, false>{v}, t2, l2);
            __syncthreads();
            F::template gather<P::R3>(PartRef<cpx<T>, false>{v}, t2, l2);
            __syncthreads();
            F::template scatter<P::R2, 16>(PartRef<cpx<T>, true>{v}, t2, l2);
            __syncthreads();
            F::template gather<P::R3>(PartRef<cpx<T>, true>{v}, t2, l2);
        } else {
            F::template scatter<P::R2, 16>(v, t2, l2);
            __syncthreads();
            F::template gather<P::R3>(v, t2, l2);
        }
        F::template compute<P::R3, 16 * P::R2, DIR>(v, t2, tw);
    }

    // ---- autosorted store
    constexpr int RL = P::R3 > 1 ? P::R3 : P::R2;
    constexpr int NSL = RP / RL;
    const size_t jj = j0 + c2;
    if constexpr (TL == 1) {
        const int lw2 = aux;
        const size_t ncols = n / RP;
        cpx<T>* out = dst + vec * n;
#pragma unroll
        for (int b = 0; b < 16 / RL; ++b)
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const unsigned k1 = (unsigned)F::template out_index<RL, NSL>(t2, b, r);
                out[((((size_t)(k1 >> lw2)) * ncols + jj) << lw2) + (k1 & ((1u << lw2) - 1u))] = v[b * RL + r];
            }
        return;
    }
    const size_t base = (jj / nsg) * nsg * RP + (jj % nsg);
    bool staged = false;
    if constexpr (GEN && !ROWMAP) {
        if (last) {
            staged = true;
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 16 / RL; ++b)
#pragma unroll
                for (int r = 0; r < RL; ++r)
                    l2[F::pad(F::template out_index<RL, NSL>(t2, b, r))] = v[b * RL + r];
            __syncthreads();
#pragma unroll 1
            for (int e = 0; e < 16; ++e) {
                int row = t2 + e * NT;
                io_store(io, vec, base + (size_t)row * nsg, l2[F::pad(row)]);
            }
        }
    }
    if (!staged) {
        cpx<T>* out = dst + vec * n + base;
        // last pass: fft_shift = the row index's top bit flipped; the last inner stage's digit r is that top digit
        const int sx = (!SIMPLE && last && !ROWMAP && (io.flags & BDSP_FFT_SHIFT_OUT)) ? RL / 2 : 0;
        if (!SIMPLE && last && !ROWMAP && (io.flags & FFT_WINDOW_OUT_DIV) && io.window_id >= 0 && io.window_id <= 2) {
            // windowed_ifft: the division by the window in the registers of the last pass (round 4; before: the staged
            // loop, 16M points f32 150 us for Hamming and 181 for Blackman-Harris against 133 for ifft).  Register
            // b RL + r goes to row t2 + (b + (r ^ sx) 16/RL) RP/16, i.e. lattice point b + (r ^ sx) 16/RL of jj + t2 n/RP
            pass_window16<T, (SPLIT ? (WIN >= 0 ? WIN : 1) : -1), true>(io, jj + (size_t)t2 * nsg, n, v, [&](int reg) { return reg / RL + ((reg % RL) ^ sx) * (16 / RL); });
        }
        if (!SIMPLE && last && !ROWMAP && (io.flags & (BDSP_FFT_MAGNITUDE | FFT_OUT_REAL))) {
            // magnitude / real part straight from the registers: `points` reals per vector
            T* outr = reinterpret_cast<T*>(io.out) + vec * io.out_stride + base;
            const bool mag = (io.flags & BDSP_FFT_MAGNITUDE) != 0;
#pragma unroll
            for (int b = 0; b < 16 / RL; ++b)
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const cpx<T> z = v[b * RL + r];
                    // (streamed with non-temporal stores: C2 x 64 377 -> 422 us -- 4-byte pieces of a line arrive from different
                    // lanes and instructions and need the cache to be merged)
                    outr[(size_t)F::template out_index<RL, NSL>(t2, b, r ^ sx) * nsg] = mag ? dev_hypot<T>(z.x, z.y) : z.x;
                }
            return;
        }
#pragma unroll
        for (int b = 0; b < 16 / RL; ++b)
#pragma unroll
            for (int r = 0; r < RL; ++r)
                // (non-temporal stores here, -DBDSP_FFT_NT in the lab build: 16M f32 points 127 -> 181 us, C2 x 64 379 -> 427
                // -- a pass's output is the next pass's input and the Infinity Cache holds it; 2^25 / 2^26 points and 16M f64,
                // whose buffers exceed the cache: +-1 %; streaming only the LAST pass's result: no difference.  Not adopted.)
#if defined(BDSP_LAB) && defined(BDSP_FFT_NT)
                nt_store(&out[(size_t)F::template out_index<RL, NSL>(t2, b, r ^ sx) * nsg], v[b * RL + r]);
#else
                out[(size_t)F::template out_index<RL, NSL>(t2, b, r ^ sx) * nsg] = v[b * RL + r];
#endif
    }
}


// ------------------------------------------------------------------------------ one 2^20-point vector (round 4, LAB only)
// *Measured and NOT adopted* (profiles/r04_c2_tile_geometry.txt): correct for every output option (rel-L2 2.1e-7) and
// 19.15 us against 17.2 us for the 1024 x 4 tiles -- the chain of a tile is memory latency, barriers and the launch
// boundary, not instruction issue, so twice the waves at half the instructions each buy nothing and the doubled L2 -> CU
// traffic costs.  Kept in the LAB build (BDSP_FFT_H512=1) so that the measurement can be repeated.
// A single 1M-point f32 transform (BASELINE config C2) is two dependent launches of 256 tiles of 1024 x 4 points: ONE wave
// per SIMD, nothing to hide the load -> butterflies -> store chain behind (9.1 + 8.0 us, of which ~1.7 us each are the
// launch boundary; 17 MB in 9 us = 1.8 TB/s).  Here every 1024-point column transform is split in two by one
// decimation-in-frequency step done in registers on load,
//     X[2k]   = FFT_512(a + b)[k]                 a[n] = x[n], b[n] = x[n + 512]
//     X[2k+1] = FFT_512((a - b) w_1024^n)[k]
// and the halves go to DIFFERENT workgroups: 512 workgroups of 256 threads x 8 points (512 = 8 x 8 x 8: two LDS
// exchanges as before), two per CU = two waves per SIMD, about half the instructions per wave; both workgroups of a tile
// load all 1024 rows (the second read is an L2 hit).  Pass 1 writes half h of column j to mid[j 1024 + 512 h + k] (whole
// contiguous runs; the parity interleave would be 8-byte pieces), so pass 2 finds output column 2k + h of pass 1 at
// column position 512 h + k: its 4-wide tile takes positions {2t, 2t+1, 512 + 2t, 512 + 2t + 1} = the four ADJACENT true
// columns 4t + {0, 2, 1, 3}.  Pass 2's outputs are rows 2 k' + h', 8 KB apart anyway.
#ifdef BDSP_LAB
template <int DIR, bool FIRST>
__global__ __launch_bounds__(256) void k_fft_half512(FftIo<float> io, const cpx<float>* __restrict__ src, cpx<float>* __restrict__ dst,
                                                      const cpx<float>* __restrict__ wtab /* exp(-2 pi i m / 1024) */)
{
    using T = float;
    using C_ = cpx<T>;
    constexpr int RP = 1024, H = 512, W = 4, NT = 64;
    constexpr int CS = col_stride(H, W);
    constexpr size_t N = (size_t)RP * RP;
    using F = WgFft<T, H, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C_* lds = reinterpret_cast<C_*>(smem_raw);
    C_* ltw = lds + (size_t)W * CS; // the 1024-entry table
    const int tid = threadIdx.x;
    for (int i = tid; i < RP; i += 256) ltw[i] = wtab[i];
    const unsigned tiles = RP / W; // 256
    unsigned tile = blockIdx.x % tiles;
    const unsigned h = blockIdx.x / tiles; // (blocks b and b + 256 sit on the same XCD: 256 % 8 == 0)
    tile = (tile & 7) * (tiles >> 3) + (tile >> 3);
    auto tw = [&](int m) { return ltw[2 * m]; }; // exp(-2 pi i m / 512)
    const int c = tid % W, ti = tid / W;
    C_ a[8], b[8];
    unsigned kt = 0; // pass 2: the true column index of this lane's column
    if (FIRST) {
        const C_* in = src + (size_t)W * tile + c;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            a[r] = in[(size_t)(ti + 64 * r) * RP];
            b[r] = in[(size_t)(ti + 64 * r + H) * RP];
        }
    } else {
        const unsigned hh = c >> 1, kk = c & 1, pos = hh * H + 2 * tile + kk;
        kt = 2 * (2 * tile + kk) + hh;
        const C_* in = src + pos;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            a[r] = in[(size_t)(ti + 64 * r) * RP];
            b[r] = in[(size_t)(ti + 64 * r + H) * RP];
        }
    }
    __syncthreads(); // the table is in LDS
    C_ v[8];
    if (FIRST) {
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = cadd(a[r], b[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = twmul<DIR>(csub(a[r], b[r]), ltw[ti + 64 * r]);
        }
    } else {
        // inter-pass twiddle w_N^(n kt) on row n = ti + 64 r (+ 512 for b): a_r P_r and b_r P_r Q with P_r = bs d^r,
        // bs = w_N^(ti kt), d = w_N^(64 kt), Q = w_N^(512 kt); the odd half's w_1024^n = w_N^(1024 n) joins bs and d
        const unsigned e1 = h ? 1024u : 0u;
        const C_ bs = unit_root<T>(((size_t)ti * (kt + e1)) & (N - 1), N);
        const C_ d1 = unit_root<T>(((size_t)64 * (kt + e1)) & (N - 1), N);
        const C_ q = unit_root<T>(((size_t)512 * kt) & (N - 1), N);
        const C_ d2 = cmul(d1, d1), d4 = cmul(d2, d2);
        C_ pw[8];
        pw[0] = bs; pw[1] = cmul(bs, d1); pw[2] = cmul(bs, d2); pw[3] = cmul(pw[1], d2);
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[4 + r] = cmul(pw[r], d4);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const C_ tb = twmul<DIR>(b[r], q);
            v[r] = twmul<DIR>(h ? csub(a[r], tb) : cadd(a[r], tb), pw[r]);
        }
    }
    // ---- 512-point transform of every column: 8 x 8 x 8
    F::template compute<8, 1, DIR>(v, ti, tw);
    const int c2 = FIRST ? tid / NT : c, t2 = FIRST ? tid % NT : ti; // pass 1: lanes along rows for the contiguous store
    C_* l1 = lds + (size_t)c * CS;
    C_* l2 = lds + (size_t)c2 * CS;
    F::template scatter<8, 1>(v, ti, l1);
    __syncthreads();
    F::template gather<8>(v, t2, l2);
    F::template compute<8, 8, DIR>(v, t2, tw);
    __syncthreads();
    F::template scatter<8, 8>(v, t2, l2);
    __syncthreads();
    F::template gather<8>(v, t2, l2);
    F::template compute<8, 64, DIR>(v, t2, tw);
    // v[r] = Y_h[t2 + 64 r]
    if (FIRST) {
        C_* out = dst + ((size_t)W * tile + c2) * RP + (size_t)h * H + t2;
#pragma unroll
        for (int r = 0; r < 8; ++r) out[64 * r] = v[r];
    } else {
        // true output row 2 k' + h; fft_shift = the row index's top bit flipped
        const unsigned sx = (io.flags & BDSP_FFT_SHIFT_OUT) ? (unsigned)H : 0u;
        if (io.flags & (BDSP_FFT_MAGNITUDE | FFT_OUT_REAL)) {
            T* outr = reinterpret_cast<T*>(io.out) + kt;
            const bool mag = (io.flags & BDSP_FFT_MAGNITUDE) != 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned row = (2u * (t2 + 64 * r) + h) ^ sx;
                outr[(size_t)row * RP] = mag ? dev_hypot<T>(v[r].x, v[r].y) : v[r].x;
            }
        } else {
            C_* out = dst + kt;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned row = (2u * (t2 + 64 * r) + h) ^ sx;
                out[(size_t)row * RP] = v[r];
            }
        }
    }
}
#endif // BDSP_LAB

// ------------------------------------------------------------------------------ launchers
// plain complex in/out with the natural batch stride and no fused option?  The input side and the
// output side are judged separately so that e.g. fft->magnitude pays the staged path only in its
// last pass.
template <typename T>
static bool io_in_generic(const FftIo<T>& io)
{
    // ifft_shift (a register renaming for even n) and the input scale are handled by the plain path
    // ... and so is a generalised Hamming window on the first global pass (n > 4096)
    // (... and so are all four reference windows -- the rectangular one multiplies by one -- on the first global pass)
    const bool win = io.window_id >= 0 && !(io.flags & FFT_WINDOW_OUT_DIV) && !(io.window_id <= 3 && io.n > 4096);
    // ... and real input (zero imaginary parts)
    return win || io.in_stride != io.n || (io.in_valid != 0 && io.n > 4096);
}
template <typename T>
static bool io_out_generic(const FftIo<T>& io)
{
    // fft_shift is handled by the plain path
    // ... and magnitude / real-part outputs are written straight from the registers
    // ... and so is the division by one of the reference's windows (windowed_ifft) on the last global pass
    return ((io.flags & FFT_WINDOW_OUT_DIV) != 0 && !(io.window_id >= 0 && io.window_id <= 3 && io.n > 4096)) || io.out_stride != io.n;
}
template <typename T>
static bool io_is_generic(const FftIo<T>& io) { return io_in_generic(io) || io_out_generic(io); }

template <typename T>
static size_t wg_lds_bytes(int n)
{
    int nt = n / 16;
    int b = 256 / nt;
    return (size_t)b * col_stride(n) * sizeof(cpx<T>);
}

template <typename K>
static int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return BDSP_OK;
}

template <typename T, int N>
static int launch_wg4(const FftIo<T>& io, size_t batch, bool inverse, hipStream_t s)
{
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(N, &wtab));
    using F = WgFft<T, N, N / 16>;
    const size_t lds = (size_t)(F::LDS_ELEMS + 16 * 17) * sizeof(cpx<T>);
    const size_t slots = (size_t)num_cus() * (N == 8192 ? 2 : 1);
    const unsigned grid = (unsigned)(batch < slots ? batch : slots);
    if (inverse) {
        BDSP_TRY(set_lds(k_fft_wg4<T, N, 1>, lds));
        hipLaunchKernelGGL((k_fft_wg4<T, N, 1>), dim3(grid), dim3(N / 16), lds, s, io, wtab, batch);
    } else {
        BDSP_TRY(set_lds(k_fft_wg4<T, N, -1>, lds));
        hipLaunchKernelGGL((k_fft_wg4<T, N, -1>), dim3(grid), dim3(N / 16), lds, s, io, wtab, batch);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template <typename T, int N>
static int launch_wg(const FftIo<T>& io, size_t batch, bool inverse, hipStream_t s)
{
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(N, &wtab));
    constexpr int B = 256 / (N / 16);
    size_t lds = wg_lds_bytes<T>(N);
    unsigned grid = (unsigned)((batch + B - 1) / B);
    const bool gen = io_is_generic(io);
    // k_fft_wg_batch (persistent workgroups) serves f32 only.  Round 5, on valid data (profiles/r05_plan_probe_valid.txt; the
    // round-2 figures came from loops on inf / NaN): f32 4096 x 4096 points 53.5 us cold / 46 hot against 61.3 / 49.7 for the
    // plain launch, 16384 x 4096 213 / 190 against 209-220 / 200-233; f64 LOSES at every size -- 16384 x 1024 points 107 / 94
    // against 97 / 89, x 2048 210 / 199 against 186 / 171, x 4096 439 / 426 against 387 / 374, 4096 x 4096 113 / 97 against
    // 105 / 96 -- so its f64 instantiations exist in the LAB build only (BDSP_FFT_WGBATCH_F64).
#ifdef BDSP_LAB
    static const bool wgbatch_f64 = lab_flag("BDSP_FFT_WGBATCH_F64");
    constexpr bool WGBATCH = N >= 1024;
    const bool wgbatch_on = sizeof(T) == 4 || wgbatch_f64;
#else
    constexpr bool WGBATCH = N >= 1024 && sizeof(T) == 4;
    const bool wgbatch_on = true;
#endif
#if BDSP_FFT_PART == 1
    if (gen) return launch_wg_gen<T>(N, io, batch, inverse, s); // the options unit
#endif
    if constexpr (WGBATCH && BDSP_FFT_PART != 2) {
        // enough transforms to keep persistent workgroups busy for several rounds
        // resident workgroups per CU as the runtime computes it (registers and LDS), once per kernel
        size_t lds2 = lds + 16 * 17 * sizeof(cpx<T>);
        static int occ = 0;
        if (occ == 0) {
            int o = 0;
            (void)set_lds(k_fft_wg_batch<T, N, -1>, lds2);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_fft_wg_batch<T, N, -1>, 256, lds2) != hipSuccess || o < 1) o = 1;
            occ = o;
        }
        const size_t slots = (size_t)num_cus() * (size_t)occ;
        static const bool no_wgbatch = lab_flag("BDSP_FFT_NO_WGBATCH");
        if (wgbatch_on && !no_wgbatch && !gen && grid >= 4 * slots && !(io.flags & (BDSP_FFT_MAGNITUDE | FFT_OUT_REAL | FFT_IN_REAL))) {
            if (inverse) {
                BDSP_TRY(set_lds(k_fft_wg_batch<T, N, 1>, lds2));
                hipLaunchKernelGGL((k_fft_wg_batch<T, N, 1>), dim3((unsigned)slots), dim3(256), lds2, s, io, wtab, batch);
            } else {
                BDSP_TRY(set_lds(k_fft_wg_batch<T, N, -1>, lds2));
                hipLaunchKernelGGL((k_fft_wg_batch<T, N, -1>), dim3((unsigned)slots), dim3(256), lds2, s, io, wtab, batch);
            }
            BDSP_LAUNCH_CHECK();
            return BDSP_OK;
        }
    }
#define BDSP_WG(DIRV, GENV)                                                                        \
    do {                                                                                           \
        BDSP_TRY(set_lds(k_fft_wg<T, N, DIRV, GENV>, lds));                                        \
        hipLaunchKernelGGL((k_fft_wg<T, N, DIRV, GENV>), dim3(grid), dim3(256), lds, s, io, wtab,  \
                           batch);                                                                 \
    } while (0)
#if BDSP_FFT_PART == 1
    if (inverse) BDSP_WG(1, false); else BDSP_WG(-1, false);
#elif BDSP_FFT_PART == 2
    if (!gen) { set_last_error("plain transform in the options unit"); return BDSP_ERR_UNSUPPORTED; }
    if (inverse) BDSP_WG(1, true); else BDSP_WG(-1, true);
#else
    if (inverse) { if (gen) BDSP_WG(1, true); else BDSP_WG(1, false); }
    else { if (gen) BDSP_WG(-1, true); else BDSP_WG(-1, false); }
#endif
#undef BDSP_WG
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template <typename T, int N>
static int launch_tiny(const FftIo<T>& io, size_t batch, bool inverse, hipStream_t s)
{
    unsigned grid = (unsigned)((batch + 255) / 256);
    if (inverse) hipLaunchKernelGGL((k_fft_tiny<T, N, 1>), dim3(grid), dim3(256), 0, s, io, batch);
    else hipLaunchKernelGGL((k_fft_tiny<T, N, -1>), dim3(grid), dim3(256), 0, s, io, batch);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// column lengths whose pass kernels also exist with the tiled intermediate (TL): the long ones, whose tiles are narrow --
// in the LAB build only (*measured*, round 4: not faster, see fft_pow2)
template <int RP>
constexpr bool pass_tiled_pair()
{
#ifdef BDSP_LAB
    return RP >= 512;
#else
    return false;
#endif
}

// first-pass tiles that read their (dead) input with non-temporal loads: f64, whole 128-byte lines per row (launch_pass)
template <typename T, int RP, int W>
constexpr bool pass_ntl_candidate()
{
    return sizeof(T) == 8 && (size_t)W * sizeof(cpx<T>) >= 128;
}

// tiles the product's plans use as a FIRST pass only: their later-pass instantiations are not built (LAB: all are)
template <int RP, int W>
constexpr bool pass_first_only()
{
#ifdef BDSP_LAB
    return false;
#else
    return RP == 4096;
#endif
}

template <typename T, int RP, int W>
static int launch_pass(const FftIo<T>& io_in, const cpx<T>* src, cpx<T>* dst, size_t n, size_t nsg,
                       size_t batch, bool inverse, bool first, bool last, hipStream_t s, int tl = 0, int aux = 0)
{
    FftIo<T> io = io_in;
    const cpx<T>* const src0 = src;
    cpx<T>* const dst0 = dst;
    (void)src0; (void)dst0; // (used by the plain unit only, BDSP_FFT_PART == 1)
    if ((first || last) && (io.window_id == 1 || io.window_id == 2) && n > 1) {
        // the window constants of k_fft_pass: cos / sin of q * 2 pi (n/16) / (n-1), q = 0..7, from double precision
        const double step = 2.0 * (double)(n / 16) / ((double)n - 1.0); // in units of pi
        for (int q = 0; q < 8; ++q) {
            io.win_c[q] = (T)std::cos(M_PI * step * q);
            io.win_s[q] = (T)std::sin(M_PI * step * q);
        }
    }
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(RP, &wtab));
    constexpr int THREADS = W * (RP / 16);
    size_t tiles = (n / RP) / W;
    dim3 grid((unsigned)(tiles * batch));
    const bool rowmap = nsg == 1; // the first pass
    const bool gen = (first && io_in_generic(io)) || (last && io_out_generic(io));
    // (triangular / Blackman-Harris on a split-exchange f64 tile: the kernel instantiated for that window, see k_fft_pass)
    const bool win_here = io.window_id >= 0 && ((first && !(io.flags & FFT_WINDOW_OUT_DIV)) || (last && (io.flags & FFT_WINDOW_OUT_DIV)));
    const int win_fixed = (!gen && win_here && pass_split_exchange<T, RP, W, false>() && (io.window_id == 0 || io.window_id == 2)) ? io.window_id : -1;
    (void)win_fixed; // (the plain unit sends every windowed pass to the options unit)
    if (first && !gen) src = reinterpret_cast<const cpx<T>*>(io.in);
    if (last && !gen) dst = reinterpret_cast<cpx<T>*>(io.out);
    // plain first / last pass?  (then no option is looked at inside the kernel)
    const bool simple = !gen && (rowmap ? ((io.flags & (FFT_IN_REAL | BDSP_FFT_SHIFT_IN)) == 0 && io.in_scale == (T)1 && io.window_id < 0)
                                        : (!last || (io.flags & (BDSP_FFT_SHIFT_OUT | BDSP_FFT_MAGNITUDE | FFT_OUT_REAL | FFT_WINDOW_OUT_DIV)) == 0));
#if BDSP_FFT_PART == 1
    if (!simple) return launch_pass_opts_rp<T>(RP, W, io_in, src0, dst0, n, nsg, batch, inverse, first, last, s, tl, aux); // the options unit
#endif
    // Non-temporal loads in the FIRST pass of an f64 transform whose tile reads whole 128-byte lines (W >= 8: the 3-pass plans'
    // 256 x 16 / 128 x 32 tiles, the 1024 x 8 / 512 x 8 tiles of batches and of 2^18 points).  *Measured* (round 5, LAB build with
    // -DBDSP_FFT_NTLOAD=2, cold / input in the caches, profiles/r05_plan_probe_valid.txt runs 4-5): 16 x 2^20 214 -> 185 / 205 -> 185 us,
    // 4 x 2^20 61.6 -> 54.0 / =, 2^23 141 -> 133 / 130 -> 127, 2^24 327 -> 324 / 312 -> 304, 32 x 2^20 419 -> 417 / 404 -> 398, 256 x 2^16 and
    // 2^18 equal.  NOT where the tile reads half lines (2^21 / 2^22 f64 single vectors: cold 36.6 -> 33.7 / 67.2 -> 60.8 but with
    // the input in the caches 28.1 -> 32.5 / 46.1 -> 62.3 -- every half line becomes a fetch of its own), and not in f32, where it
    // is a wash or worse (2^23 -5 % cold, 2^24 -6 % cold / +1 % hot, 2^25 +6 %, 64 x 2^20 +8 %, 4096 x 2^14 +1 %).
    const bool ntl = pass_ntl_candidate<T, RP, W>() && first && rowmap && !gen && !(io.flags & FFT_IN_REAL);
    // (tiled instantiations exist for the long columns only: pass_tiled_pair)
    if (tl != 0 && (!pass_tiled_pair<RP>() || (tl == 1) != rowmap)) { set_last_error("tiled intermediate: unsupported pass"); return BDSP_ERR_UNSUPPORTED; }
#define BDSP_PASS_K(DIRV, RM, GENV, SV, TLV, WINV, NTLV)                                            \
    do {                                                                                           \
        constexpr size_t lds = pass_tile_lds_bytes<T, RP, W, GENV>() +                             \
                               (pass_lds_twiddles<T, RP, W, GENV>() ? (size_t)RP * sizeof(cpx<T>) : 0); \
        BDSP_TRY(set_lds(k_fft_pass<T, RP, W, DIRV, RM, GENV, SV, TLV, WINV, NTLV>, lds));         \
        hipLaunchKernelGGL((k_fft_pass<T, RP, W, DIRV, RM, GENV, SV, TLV, WINV, NTLV>), grid, dim3(THREADS), lds, s, \
                           io, src, dst, wtab, n, nsg, tiles, (int)last, aux);                     \
    } while (0)
#define BDSP_PASS(DIRV, RM, GENV, SV, TLV, WINV)                                                   \
    do {                                                                                           \
        if constexpr (pass_ntl_candidate<T, RP, W>() && RM && !GENV && TLV == 0) {                 \
            if (ntl) { BDSP_PASS_K(DIRV, RM, GENV, SV, TLV, WINV, true); break; }                  \
        }                                                                                          \
        BDSP_PASS_K(DIRV, RM, GENV, SV, TLV, WINV, false);                                         \
    } while (0)
#if BDSP_FFT_PART == 1
#define BDSP_PASS_V(DIRV, RM, TLV) BDSP_PASS(DIRV, RM, false, true, TLV, -1) /* (everything else went to the options unit above) */
#else
#if BDSP_FFT_PART == 2
#define BDSP_PASS_SIMPLE(DIRV, RM, TLV) do { set_last_error("plain pass in the options unit"); return BDSP_ERR_UNSUPPORTED; } while (0)
#else
#define BDSP_PASS_SIMPLE(DIRV, RM, TLV) BDSP_PASS(DIRV, RM, false, true, TLV, -1)
#endif
#define BDSP_PASS_V(DIRV, RM, TLV)                                                                 \
    do {                                                                                           \
        if (gen) BDSP_PASS(DIRV, RM, true, false, TLV, -1);                                        \
        else if (simple) BDSP_PASS_SIMPLE(DIRV, RM, TLV);                                          \
        else {                                                                                     \
            if constexpr (pass_split_exchange<T, RP, W, false>() && TLV == 0) {                    \
                if (win_fixed == 0) { BDSP_PASS(DIRV, RM, false, false, TLV, 0); break; }          \
                if (win_fixed == 2) { BDSP_PASS(DIRV, RM, false, false, TLV, 2); break; }          \
            }                                                                                      \
            BDSP_PASS(DIRV, RM, false, false, TLV, -1);                                            \
        }                                                                                          \
    } while (0)
#endif
#define BDSP_PASS_D(DIRV)                                                                          \
    do {                                                                                           \
        if constexpr (pass_tiled_pair<RP>()) {                                                     \
            if (tl == 1) { BDSP_PASS_V(DIRV, true, 1); break; }                                    \
            if (tl == 2) { BDSP_PASS_V(DIRV, false, 2); break; }                                   \
        }                                                                                          \
        if (rowmap) BDSP_PASS_V(DIRV, true, 0);                                                    \
        else if constexpr (pass_first_only<RP, W>()) { set_last_error("first-pass-only tile"); return BDSP_ERR_UNSUPPORTED; } \
        else BDSP_PASS_V(DIRV, false, 0);                                                          \
    } while (0)
    if (inverse) BDSP_PASS_D(1); else BDSP_PASS_D(-1);
#undef BDSP_PASS_D
#undef BDSP_PASS_V
#undef BDSP_PASS
#undef BDSP_PASS_K
#if BDSP_FFT_PART != 1
#undef BDSP_PASS_SIMPLE
#endif
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// the two launches of k_fft_half512 for ONE plain 2^20-point f32 vector (io.in -> scratch_a -> io.out)
#if defined(BDSP_FFT_F32_TU) && defined(BDSP_LAB)
static int launch_half512(const FftIo<float>& io, cpx<float>* mid, bool inverse, hipStream_t s)
{
    const cpx<float>* wtab;
    BDSP_TRY(twiddle_table<float>(1024, &wtab));
    constexpr size_t lds = ((size_t)4 * col_stride(512, 4) + 1024) * sizeof(cpx<float>);
    const cpx<float>* in = reinterpret_cast<const cpx<float>*>(io.in);
    cpx<float>* out = reinterpret_cast<cpx<float>*>(io.out);
    if (inverse) {
        hipLaunchKernelGGL((k_fft_half512<1, true>), dim3(512), dim3(256), lds, s, io, in, mid, wtab);
        hipLaunchKernelGGL((k_fft_half512<1, false>), dim3(512), dim3(256), lds, s, io, mid, out, wtab);
    } else {
        hipLaunchKernelGGL((k_fft_half512<-1, true>), dim3(512), dim3(256), lds, s, io, in, mid, wtab);
        hipLaunchKernelGGL((k_fft_half512<-1, false>), dim3(512), dim3(256), lds, s, io, mid, out, wtab);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}
#else
template <typename T>
static int launch_half512(const FftIo<T>&, cpx<T>*, bool, hipStream_t) { return BDSP_ERR_UNSUPPORTED; } // (LAB build, f32 TU only)
#endif

// (super-radix, tile width) pairs that are instantiated.  Default tile: W = 4096/RP columns
// (256 threads); the wider variants trade LDS for longer contiguous global segments.
template <typename T>
static int launch_pass_rp(int rp, int w, const FftIo<T>& io, const cpx<T>* src, cpx<T>* dst,
                          size_t n, size_t nsg, size_t batch, bool inverse, bool first, bool last,
                          hipStream_t s, int tl = 0, int aux = 0)
{
#define BDSP_CASE(RPV, WV)                                                                         \
    if (rp == RPV && w == WV)                                                                      \
        return launch_pass<T, RPV, WV>(io, src, dst, n, nsg, batch, inverse, first, last, s, tl, aux);
    // every pair plan_passes can choose by itself ...
    BDSP_CASE(64, 64) BDSP_CASE(128, 32) BDSP_CASE(256, 16) BDSP_CASE(512, 8) BDSP_CASE(1024, 4)
    BDSP_CASE(1024, 8) BDSP_CASE(2048, 4)
    // (the first pass of 2^22 f32 points, round 5: the ROWMAP instantiations only, see pass_first_only)
    if constexpr (sizeof(T) == 4) { BDSP_CASE(4096, 4) }
#ifdef BDSP_LAB
    // ... and the ones only a BDSP_FFT_PLAN experiment reaches (the f64 4096 x 4 tiles spill 12-76 bytes per lane)
    BDSP_CASE(2048, 2) BDSP_CASE(4096, 2) BDSP_CASE(4096, 4) BDSP_CASE(1024, 2)
#endif
#undef BDSP_CASE
    set_last_error("unsupported super-radix / tile width");
    return BDSP_ERR_UNSUPPORTED;
}

// Super-radix plan for n = 2^bits > 4096: 2 passes up to 2^20, 3 passes up to 2^30, bits split as
// evenly as possible, largest first (the first pass is the one whose stores are always long
// contiguous runs, so it can afford the narrowest tile).
static int plan_passes(size_t n, size_t batch, size_t esz, int rp[3], int w[3])
{
    int bits = 0;
    while ((size_t(1) << bits) < n) ++bits;
    if (bits > 30) return 0;
    // experiment hook (lab build only): BDSP_FFT_PLAN="4096x2,4096x2" (radix x tile width per pass)
    if (const char* e = lab_env("BDSP_FFT_PLAN")) {
        int k = 0;
        size_t prod = 1;
        while (*e && k < 3) {
            int r = 0, ww = 0;
            if (sscanf(e, "%dx%d", &r, &ww) != 2) break;
            rp[k] = r; w[k] = ww; prod *= (size_t)r; ++k;
            while (*e && *e != ',') ++e;
            if (*e == ',') ++e;
        }
        if (k >= 2 && prod == n) return k;
    }
    // 2^21 and 2^22 points: two passes of long columns beat three fully coalesced ones although their runs are only 32-64
    // bytes.  Round 5, re-measured on VALID data with input and scratch cold / input in the caches (tools/plan_probe.py,
    // profiles/r05_plan_probe_valid.txt; the figures of rounds 2-3 came from loops that transformed their own output, i.e.
    // inf / NaN), last pass in place:
    //   2^21 f32  1024x8 + 2048x4  24.3 / 19.8 us   (2048x4 + 1024x8 24.6 / 19.2; three passes 30.3 / 26.9 out of place)
    //   2^22 f32  4096x4 + 1024x8  40.4 / 32.3      (2048x4 + 2048x4, the plan until round 5: 42.9 / 33.3; three passes 41.4 / 34.0)
    //   2^21 f64  1024x4 + 2048x4  36.6 / 28.4      (three passes 39.5 / 34.5)
    //   2^22 f64  2048x4 + 2048x4  66.9 / 47.0      (2048x2 + 2048x4 70.8 / 50.8; three passes 76.5 / 65.6)
    // From 2^23 on three passes win on a cold input: f32 71.2 / 65.7 against 78.9-86.7 / 63.3-63.8 for the 4096-point
    // columns, f64 137.9 / 131.1 against 140-147 / 121-132; 2^24 f32 132.5 / 129.2 against 173 / 149.
    if (bits == 21 || bits == 22) {
        if (esz == 4) {
            if (bits == 21) { rp[0] = 1024; w[0] = 8; rp[1] = 2048; w[1] = 4; }
            else { rp[0] = 4096; w[0] = 4; rp[1] = 1024; w[1] = 8; }
        } else {
            rp[0] = bits == 21 ? 1024 : 2048; rp[1] = 2048;
            w[0] = 4; w[1] = 4; // 2^22 f64: 2-wide first-pass tiles are no faster plain (above) and 7 us slower with a fused window
        }
        return 2;
    }
    int passes = bits <= 20 ? 2 : 3;
    int base = bits / passes, extra = bits % passes;
    for (int i = 0; i < passes; ++i) {
        rp[i] = 1 << (base + (i < extra ? 1 : 0));
        w[i] = 4096 / rp[i];
        // 1024-point columns: 8-wide tiles (64-byte segments, 512 threads) measured 14 % faster than
        // 4-wide ones on 64 x 2^20 points; 2-pass plans for 2^24 (4096x2 / 4096x4 tiles) measured
        // 30-70 % SLOWER than three fully coalesced 256-point passes, so 3 passes stay.
        // ... unless that leaves fewer tiles than two per CU (a single 2^20-point vector has 128):
        // then 4-wide tiles fill the chip (measured 19.8 -> 17.2 us for one 2^20-point transform)
        if (rp[i] == 1024) w[i] = (n * batch / 8192 >= (size_t)2 * num_cus()) ? 8 : 4;
    }
    return passes;
}

// lengths whose PLAIN transform (no fused option) runs in the one-workgroup kernel k_fft_wg4 instead of two passes
template <typename T>
static bool wg4_serves(size_t n)
{
    static const bool no_wg4 = lab_flag("BDSP_FFT_NO_WG4");
    return sizeof(T) == 4 && n == 8192 && !no_wg4;
}

// Runs the transform described by `io` on power-of-two n.  io.in holds the input.  For n <= 4096
// the result goes to io.out (io.out may equal io.in unless the input is real).  For n > 4096 the
// passes ping-pong: in -> scratch_a -> [scratch_b ->] io.out; scratch buffers hold n*batch complex.
// scratch_b may be null for 2-pass sizes, and may alias io.in if the caller allows the input to be
// clobbered; io.out may alias io.in (it is written only by the last pass, which reads scratch).
template <typename T>
int fft_pow2(const FftIo<T>& io, T* scratch_a, T* scratch_b, size_t batch, bool inverse,
             hipStream_t s)
{
    const size_t n = io.n;
    if (batch == 0 || n == 0) return BDSP_OK;
    if (n == 1) {
        // a 1-point DFT is the identity; still honour scale/window/magnitude through the tiny path
        return launch_tiny<T, 1>(io, batch, inverse, s);
    }
    switch (n) {
    case 2: return launch_tiny<T, 2>(io, batch, inverse, s);
    case 4: return launch_tiny<T, 4>(io, batch, inverse, s);
    case 8: return launch_tiny<T, 8>(io, batch, inverse, s);
    case 16: return launch_wg<T, 16>(io, batch, inverse, s);
    case 32: return launch_wg<T, 32>(io, batch, inverse, s);
    case 64: return launch_wg<T, 64>(io, batch, inverse, s);
    case 128: return launch_wg<T, 128>(io, batch, inverse, s);
    case 256: return launch_wg<T, 256>(io, batch, inverse, s);
    case 512: return launch_wg<T, 512>(io, batch, inverse, s);
    case 1024: return launch_wg<T, 1024>(io, batch, inverse, s);
    case 2048: return launch_wg<T, 2048>(io, batch, inverse, s);
    case 4096: return launch_wg<T, 4096>(io, batch, inverse, s);
    default: break;
    }
    if constexpr (sizeof(T) == 4) {
        // *measured*: 8192 points, batch 2048: 94.8 us as two passes, 73.3 us in one workgroup each (batch 256:
        // 19.8 -> 13.2 us, a single transform 7.7 vs 7.9 us); the 16384-point instantiation (1024 threads, one
        // workgroup per CU) measured SLOWER than two passes at every batch size and is not built
        // (wg4_serves: the one predicate this dispatch and bdsp_hip_fft_passes share, above)
        if (wg4_serves<T>(n) && !io_is_generic(io) && io.window_id < 0 &&
            !(io.flags & (BDSP_FFT_MAGNITUDE | FFT_OUT_REAL | FFT_IN_REAL))) return launch_wg4<T, 8192>(io, batch, inverse, s);
    }
    if constexpr (sizeof(T) == 4) {
        // ONE plain 2^20-point vector (config C2) through the half-column plan: LAB experiment only (*measured* slower,
        // round 4)
        static const bool h512 = lab_flag("BDSP_FFT_H512");
        if (h512 && n == (size_t(1) << 20) && batch == 1 && scratch_a && !io_is_generic(io) && io.window_id < 0 && io.in_scale == (T)1 &&
            !(io.flags & (FFT_IN_REAL | BDSP_FFT_SHIFT_IN)) && io.in_valid == 0 &&
            reinterpret_cast<const void*>(scratch_a) != io.in && reinterpret_cast<const void*>(scratch_a) != io.out)
            return launch_half512(io, reinterpret_cast<cpx<T>*>(scratch_a), inverse, s);
    }
    int rp[3], w[3];
    int passes = plan_passes(n, batch, sizeof(T), rp, w);
    if (passes == 0) { set_last_error("FFT length above 2^30 points"); return BDSP_ERR_UNSUPPORTED; }
    if (!scratch_a || (passes == 3 && !scratch_b)) {
        set_last_error("fft_pow2: scratch missing");
        return BDSP_ERR_UNSUPPORTED;
    }
    cpx<T>* sa = reinterpret_cast<cpx<T>*>(scratch_a);
    cpx<T>* sb = reinterpret_cast<cpx<T>*>(scratch_b);
    size_t nsg = 1;
    // two passes with narrow second-pass tiles: the intermediate through scratch_a in the TILED layout (k_fft_pass, TL) --
    // LAB experiment (BDSP_FFT_TILED=1).  *Measured*, round 4 (tools/ab_tiled.sh, profiles/r04_fft_tiled_intermediate.txt):
    // the second pass's reads become one contiguous block per tile, the first pass's stores 128-512-byte pieces instead of
    // whole 4-8 KB columns, and nothing gets faster -- C2 17.1 -> 17.5 us, C2 x 64 384 -> 409, C4a 53.2 -> 54.0, C5 equal:
    // the narrow READS were never the cost (32-byte sectors out of the Infinity Cache), and the scattered stores are one.
    static const bool want_tiled = lab_flag("BDSP_FFT_TILED");
    // (not with the triangular / Blackman-Harris window: the split-exchange f64 tiles carry those as per-window
    // instantiations, which exist for the natural-order intermediate only -- a tiled one would silently apply Hamming)
    const bool tiled = passes == 2 && want_tiled && rp[0] >= 512 && rp[1] >= 512 && (size_t)w[1] * sizeof(cpx<T>) < 128 &&
                       reinterpret_cast<const void*>(scratch_a) != io.out && io.window_id != 0 && io.window_id != 2;
    int lw2 = 0;
    while ((1 << lw2) < w[1]) ++lw2;
    BDSP_TRY(launch_pass_rp<T>(rp[0], w[0], io, nullptr, sa, n, nsg, batch, inverse, true, false, s, tiled ? 1 : 0, lw2));
    nsg *= rp[0];
    if (passes == 2) {
        BDSP_TRY(launch_pass_rp<T>(rp[1], w[1], io, sa, nullptr, n, nsg, batch, inverse, false, true, s, tiled ? 2 : 0, 0));
    } else {
        BDSP_TRY(launch_pass_rp<T>(rp[1], w[1], io, sa, sb, n, nsg, batch, inverse, false, false, s));
        nsg *= rp[1];
        BDSP_TRY(launch_pass_rp<T>(rp[2], w[2], io, sb, nullptr, n, nsg, batch, inverse, false, true, s));
    }
    return BDSP_OK;
}

// trips through global memory a power-of-two transform of n points makes (callers size and order their ping-pong
// buffers by it: 2 = in -> scratch_a -> out, 3 = in -> scratch_a -> scratch_b -> out)
template <typename T>
int fft_pow2_passes(size_t n)
{
    if (n <= 4096) return 1;
    int rp[3], w[3];
    return plan_passes(n, 1, sizeof(T), rp, w);
}

template <typename T>
int fft_pow2_plain_trips(size_t n)
{
    if (n == 0 || (n & (n - 1)) != 0) return 0;
    return wg4_serves<T>(n) ? 1 : fft_pow2_passes<T>(n);
}

#if BDSP_FFT_PART != 1
// the options unit's two entry points (with BDSP_FFT_PART == 0 they are defined here too and simply never called)
template <typename T>
int launch_pass_opts_rp(int rp, int w, const FftIo<T>& io, const cpx<T>* src, cpx<T>* dst, size_t n, size_t nsg, size_t batch,
                        bool inverse, bool first, bool last, hipStream_t s, int tl, int aux)
{
    return launch_pass_rp<T>(rp, w, io, src, dst, n, nsg, batch, inverse, first, last, s, tl, aux);
}
template <typename T>
int launch_wg_gen(int n, const FftIo<T>& io, size_t batch, bool inverse, hipStream_t s)
{
    switch (n) {
    case 16: return launch_wg<T, 16>(io, batch, inverse, s);
    case 32: return launch_wg<T, 32>(io, batch, inverse, s);
    case 64: return launch_wg<T, 64>(io, batch, inverse, s);
    case 128: return launch_wg<T, 128>(io, batch, inverse, s);
    case 256: return launch_wg<T, 256>(io, batch, inverse, s);
    case 512: return launch_wg<T, 512>(io, batch, inverse, s);
    case 1024: return launch_wg<T, 1024>(io, batch, inverse, s);
    case 2048: return launch_wg<T, 2048>(io, batch, inverse, s);
    case 4096: return launch_wg<T, 4096>(io, batch, inverse, s);
    default: break;
    }
    set_last_error("launch_wg_gen: unsupported length");
    return BDSP_ERR_UNSUPPORTED;
}
#endif
#if BDSP_FFT_PART == 2
template int launch_pass_opts_rp<BDSP_FFT_T>(int, int, const FftIo<BDSP_FFT_T>&, const cpx<BDSP_FFT_T>*, cpx<BDSP_FFT_T>*, size_t, size_t, size_t, bool, bool,
                                             bool, hipStream_t, int, int);
template int launch_wg_gen<BDSP_FFT_T>(int, const FftIo<BDSP_FFT_T>&, size_t, bool, hipStream_t);
#else
template int fft_pow2<BDSP_FFT_T>(const FftIo<BDSP_FFT_T>&, BDSP_FFT_T*, BDSP_FFT_T*, size_t, bool, hipStream_t);
template int fft_pow2_passes<BDSP_FFT_T>(size_t);
template int fft_pow2_plain_trips<BDSP_FFT_T>(size_t);
#endif

} // namespace bdsp
