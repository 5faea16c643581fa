// conv.hip -- convolve_signal on gfx950: fused overlap-save blocks (+ a direct form for the
// shapes overlap-save cannot take).
//
// Reference semantics (vector/src/vector_types/time_freq/mod.rs:455-473, convolution.rs:477-542):
//     y[i] = sum_{k=0}^{M-1} x[(i + ceil(M/2) - 1 - k) mod N] * h[k]        (centred, circular)
// Reference schedule being replaced: overlap_discard (convolution.rs:304-461) = per block
// FFT -> xH/fft_len -> IFFT with three trips through memory, a scalar O(N*M/2) tail, and on the
// OpenCL backend one context + plan + upload per call (ocl/mod.rs:361-520).
//
// Here ONE kernel does load -> FFT_L -> xH -> IFFT_L -> store of the valid part, with the block
// resident in registers/LDS the whole time: HBM sees each input sample once (plus the M-1 overlap,
// normally an L2 hit) and each output sample once = 16 B per complex f32 sample.
// Block b produces outputs [b*V, (b+1)*V), V = L-(M-1), from inputs x[(b*V - floor(M/2) + n) mod N],
// n = 0..L-1; the wrap-around makes the head and tail ordinary blocks (no scalar loops).
// L = 4096, 256 threads x 16 points, three radix-16 Stockham stages each way; the forward
// transform leaves X[t + 256 r] in register r of thread t, which is exactly what the inverse
// transform's first stage wants, so the spectrum never visits LDS.  Inner-stage twiddles live in
// registers for the whole (persistent) workgroup in the f32 build.
#include "bdsp_internal.h"
#include <type_traits>
#include <cstdlib>

namespace bdsp {

constexpr int CONV_L = 4096;

size_t conv_fft_len(size_t) { return CONV_L; }

// Grid: x = persistent workgroups walking the blocks of one vector with a grid stride,
//       y = vector of the batch.  All per-vector indices are 32-bit (points < 2^31).
//
// What bounds this kernel (MI355X measurements: an in-kernel timeline, tools/ubench/*.hip; the complex f32 case now runs
// conv_v2.hip, which documents what round 2 measured):
//   * per block and CU the butterflies need 1.39 us of VALU (620 v_pk_* instructions per wave at
//     ~4.4 clk each, saturated by ~2 waves per SIMD), the four LDS exchanges 1.16 us (64 KB each at
//     ~110 B/clk per CU, saturated by ONE workgroup), the block's share of HBM 2.0 us -- and the three
//     barely overlap: butterflies + exchanges together take 2.07 us per block with three workgroups
//     per CU, 2.32 with two, 2.40 when two subgroups are forced into anti-phase (one computes while
//     the other exchanges), i.e. ~85 % of the sum however the waves are arranged; the full kernel
//     runs at 3.75 us per block = 82 % of the sum of all three.  The HBM roofline (2.0 us) is
//     therefore out of reach for this instruction mix; what counts is instructions per block;
//   * FAST (f32): the filter spectrum H (16 values per thread) stays in registers for the whole
//     persistent loop, the stage-3 twiddles as six values (w^r = w^(4a) * w^b: nine extra multiplies
//     per butterfly, no scratch spills at 3 workgroups per CU), the stage-2 twiddles (16 distinct
//     rows) in a 2 KB LDS table (re-reading H from L2 per block cost 22 us per launch);
//   * tried and measured slower: 2 workgroups + one or two blocks of register prefetch 82 us, 4 with H
//     re-read from L2 and 20 spilled registers 99 us, ping-pong LDS buffers (4 barriers instead of 8)
//     80 us, one block per non-persistent workgroup 149 us, and a 512-thread workgroup of two
//     anti-phased subgroups with hand-scheduled memory traffic (SGPR-base addressing, staged stores,
//     explicit s_waitcnt; tools/experiments/conv_antiphase_kernel.hip.txt) 88 us;
//     and ONE WAVEFRONT per block (lane t owns x[t + 64 r]: two DFT-64 register stages per transform, two
//     LDS exchanges and no workgroup barrier at all, H and the split twiddles in the 512-register budget of a
//     single wave per SIMD; tools/experiments/conv_wave_kernel.hip.txt) 100 us -- 87 us of it with the global
//     loads removed: a lone wave per SIMD issues a dependent packed instruction only every ~11 clocks, and
//     two waves per SIMD do not fit (8 x 33 KB of exchange space, 128 registers of H per wave);
//   * two facts from that last experiment that any future prefetching version must respect: a store
//     reads its address and data VGPRs LATE, so the compiler makes arithmetic that reuses one of them
//     wait for the store to retire; and vmcnt counts loads and stores together, so with predicated
//     (variable-count) stores "wait for the prefetched loads" degenerates to "wait for everything";
//   * the f64 build keeps H (64 registers) and the split stage-3 twiddles in registers as well (240 VGPRs, 2 per CU):
//     161 us for 16M complex f64 points against 253 us with both re-read from L2 every block.
//   * REAL: the vector is REAL (and so are the taps): two consecutive blocks travel through the
//     complex transform pair as real and imaginary part (convolution with a real filter is
//     real-linear, so they come out separated) -- half the butterflies and 8 B of traffic per sample
//     instead of complexify -> convolve -> project (40 B per sample).
// Experiment switch: -DBDSP_CONV_HL2=1 re-reads H from L2 per block and builds for 4 workgroups per CU
// (128 VGPRs).  Measured 96 us against 80 us with H in registers at 3 per CU -- kept for reference.
#ifndef BDSP_CONV_HL2
#define BDSP_CONV_HL2 0
#endif
template <typename T, bool FAST, bool REAL>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? (BDSP_CONV_HL2 ? 4 : 3) : 2) void k_overlap_save(
    const void* __restrict__ x_, void* __restrict__ y_, const cpx<T>* __restrict__ hs,
    const cpx<T>* __restrict__ wtab, unsigned n, int m_taps, long long in_off, long long out_off,
    unsigned blocks_per_vec, unsigned out_limit, int store_all)
{
    constexpr int L = CONV_L, NT = 256;
    using F = WgFft<T, L, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = (unsigned)t;
    const int ov = m_taps - 1;
    // Valid outputs per block, rounded DOWN to a multiple of 16 points (128 bytes for f32): every
    // block then starts on a cache-line boundary in both the input and the output vector.  With the
    // natural V = L-(M-1) = 3073 each 512-byte wave access straddled five lines instead of four and
    // began mid-line (timeline: 12 stores took 2.3k cycles to issue).
    const unsigned Vfull = (unsigned)(L - ov);
    const unsigned V = REAL ? (Vfull >= 32 ? (Vfull & ~31u) : Vfull) : (Vfull >= 16 ? (Vfull & ~15u) : Vfull);
    const T hscale = (T)1 / (T)L; // the inverse transform below is unnormalised
    auto tw = [&](int mm) { return wtab[mm]; };

    constexpr bool HREG = !BDSP_CONV_HL2;
    // stage 3 in FMA form (fft_core.h dft16_tw, round 3): f32 holds its eight twiddle values, f64 two (w^2, w) and
    // derives the rest -- the six-value split it replaces cost nine complex multiplies per transform and 8 more registers
    // (the f64 REAL instantiation spilled 28 bytes per lane)
    constexpr bool F32 = sizeof(T) == 4;
    cpx<T> tw3f[F32 ? 8 : 2], hreg[HREG ? 16 : 1];
    cpx<T>* tw2l = lds + F::LDS_ELEMS;
    if constexpr (F32) F::template load_twiddles16_fma<256>(tw3f, t, tw);
    else { tw3f[0] = wtab[2 * t]; tw3f[1] = wtab[t]; }
    auto stage3 = [&](cpx<T> (&v)[16], auto D) {
        constexpr int DIR = decltype(D)::value;
        if constexpr (F32) dft16_tw<DIR>(&v[0], tw3f);
        else {
            cpx<T> tl[8];
            expand_twiddles16_fma<2>(tw3f, tl);
            dft16_tw<DIR>(&v[0], tl);
        }
    };
    {
        if constexpr (HREG) {
            if (!(store_all & 2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    cpx<T> hv = hs[ut + 256u * r];
                    hreg[r] = cpx<T>{hv.x * hscale, hv.y * hscale};
                }
            }
        }
    }
    // stage-2 twiddles (16 distinct rows of 15) in LDS for both builds
    if (t < 240) {
        int k = t / 15, r = t % 15 + 1;
        tw2l[k * 17 + r - 1] = wtab[r * k * 16];
    }
    __syncthreads();
    const cpx<T>* tw2p = tw2l + (t & 15) * 17;
    if constexpr (HREG) {
        // store_all bit 1: `hs` holds the TAPS, not their spectrum -- every workgroup transforms the zero-padded taps
        // itself (half a block of extra work per workgroup, in parallel) and keeps the result where the block loop wants
        // it: X[t + 256 r] in register r.  Saves the separate 7 us spectrum launch and its round trip through memory.
        if (store_all & 2) {
            cpx<T> hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned i = ut + 256u * r;
                // bit 2: the taps are REAL scalars (real signal, real filter)
                if (store_all & 4) hv[r] = i < (unsigned)m_taps ? cpx<T>{reinterpret_cast<const T*>(hs)[i], (T)0} : cpx<T>{(T)0, (T)0};
                else hv[r] = i < (unsigned)m_taps ? hs[i] : cpx<T>{(T)0, (T)0};
            }
            auto twh = [&](int mm) { return wtab[mm]; };
            F::template compute<16, 1, -1>(hv, t, twh);
            F::scatter_a(hv, t, lds);
            __syncthreads();
            F::gather_a(hv, t, lds);
            F::template compute_pre<16, 16, -1>(hv, tw2p);
            __syncthreads();
            F::scatter_b(hv, t, lds);
            __syncthreads();
            F::gather_b(hv, t, lds);
            stage3(hv, std::integral_constant<int, -1>{});
#pragma unroll
            for (int r = 0; r < 16; ++r) hreg[r] = cpx<T>{hv[r].x * hscale, hv[r].y * hscale};
            __syncthreads();
        }
    }
    const size_t vec = blockIdx.y;
    // per-vector bases; a REAL vector has n real samples (half the bytes of n complex ones)
    const cpx<T>* __restrict__ xv = REAL
        ? reinterpret_cast<const cpx<T>*>(reinterpret_cast<const T*>(x_) + vec * (size_t)n)
        : reinterpret_cast<const cpx<T>*>(x_) + vec * (size_t)n;
    cpx<T>* __restrict__ yv = REAL
        ? reinterpret_cast<cpx<T>*>(reinterpret_cast<T*>(y_) + vec * (size_t)n)
        : reinterpret_cast<cpx<T>*>(y_) + vec * (size_t)((store_all & 1) ? (unsigned)L : n);

    // block b reads x[(b*V + in_off + i) mod n], i = t + 256 r.  Global addressing is
    // uniform 64-bit base (scalar registers) + small unsigned lane index.
    // REAL: "block" b is the pair of real blocks 2b (real part) and 2b+1 (imaginary part)
    auto load_real = [&](unsigned rb, T (&d)[16]) {
        const T* __restrict__ xr = reinterpret_cast<const T*>(xv); // REAL: n real samples per vector
        long long base = (long long)rb * V + in_off;
        if (base >= 0 && base + L <= (long long)n) {
            const T* xb = xr + base;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = xb[ut + 256u * r];
        } else {
            long long sb = base % (long long)n;
            if (sb < 0) sb += n;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = xr[((unsigned long long)sb + ut + 256u * r) % n];
        }
    };
    auto load_block = [&](unsigned b, cpx<T> (&d)[16]) {
        if constexpr (REAL) {
            T re[16], im[16];
            load_real(2 * b, re);
            load_real(2 * b + 1, im); // past the end it wraps around; its outputs are never stored
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = cpx<T>{re[r], im[r]};
            return;
        }
        long long base = (long long)b * V + in_off;
        if (base >= 0 && base + L <= (long long)n) {
            const cpx<T>* xb = xv + base;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = xb[ut + 256u * r];
        } else {
            long long sb = base % (long long)n;
            if (sb < 0) sb += n;
            const unsigned idx = (unsigned)sb + ut;
            if (n >= (unsigned)L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned i = idx + 256u * r;
                    if (i >= n) i -= n;
                    d[r] = xv[i];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = xv[(idx + 256u * r) % n];
            }
        }
    };

    // Workgroup w runs on XCD w % 8 (observed dispatch order; only speed depends on it).  Blocks
    // that are adjacent in the signal share M-1 input samples, so give each XCD a contiguous run
    // of blocks per sweep: the overlap is then served by that XCD's L2 instead of HBM.
    unsigned wl = blockIdx.x;
    if ((gridDim.x & 7) == 0) wl = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    const unsigned G = gridDim.x;
    // one block: transform the 16 register-resident inputs and store the valid outputs
    auto process = [&](cpx<T> (&v)[16], unsigned b) {
        const cpx<T>* hp = hs + t;
        const cpx<T>* wt = wtab;
        if constexpr (!HREG) {
            // keep the spectrum / twiddle loads inside the loop (hoisted they would pin > 150 VGPRs)
            asm volatile("" : "+v"(hp));
            asm volatile("" : "+s"(wt));
        }
        auto twl = [&](int mm) { return wt[mm]; };

        // ---- forward FFT_L
        F::template compute<16, 1, -1>(v, t, twl);
        __syncthreads(); // previous block's last gather is done
        F::scatter_a(v, t, lds);
        __syncthreads();
        F::gather_a(v, t, lds);
        F::template compute_pre<16, 16, -1>(v, tw2p);
        __syncthreads();
        F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        stage3(v, std::integral_constant<int, -1>{});

        // ---- spectrum product
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (HREG) v[r] = cmul(v[r], hreg[r]);
            else {
                cpx<T> hv = hp[256 * r];
                v[r] = cmul(v[r], cpx<T>{hv.x * hscale, hv.y * hscale});
            }
        }

        // ---- inverse FFT_L
        F::template compute<16, 1, 1>(v, t, twl);
        __syncthreads();
        F::scatter_a(v, t, lds);
        __syncthreads();
        F::gather_a(v, t, lds);
        F::template compute_pre<16, 16, 1>(v, tw2p);
        __syncthreads();
        F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        stage3(v, std::integral_constant<int, 1>{});

        // ---- store: z[n'] for n' >= M-1 is output b*V + out_off + (n' - (M-1))
        if constexpr (REAL) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const long long obase = (long long)(2 * b + half) * V + out_off - ov;
                long long room = (long long)out_limit - obase;
                unsigned lim = room <= 0 ? 0u : (room > L ? (unsigned)L : (unsigned)room);
                if (lim > (unsigned)ov + V) lim = (unsigned)ov + V;
                T* yb = reinterpret_cast<T*>(yv) + obase;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned np = ut + 256u * r;
                    if (np >= (unsigned)ov && np < lim) yb[np] = half ? v[r].y : v[r].x;
                }
            }
        } else if (store_all & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[ut + 256u * r] = v[r];
        } else {
            // output index = obase + n', valid for ov <= n' < lim (both bounds uniform)
            const long long obase = (long long)b * V + out_off - ov;
            long long room = (long long)out_limit - obase;
            unsigned lim = room <= 0 ? 0u : (room > L ? (unsigned)L : (unsigned)room);
            if (lim > (unsigned)ov + V) lim = (unsigned)ov + V; // outputs past V belong to the next block
            cpx<T>* yb = yv + obase;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned np = ut + 256u * r;
                if (np >= (unsigned)ov && np < lim) yb[np] = v[r];
            }
        }
    };

    constexpr bool PREFETCH = false;
    if constexpr (!PREFETCH) {
        for (unsigned b = wl; b < blocks_per_vec; b += G) {
            cpx<T> v[16];
            load_block(b, v);
            process(v, b);
        }
    } else {
        cpx<T> nx[16];
        if (wl < blocks_per_vec) load_block(wl, nx);
        for (unsigned b = wl; b < blocks_per_vec; b += G) {
            cpx<T> v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = nx[r];
            if (b + G < blocks_per_vec) load_block(b + G, nx); // prefetch
            process(v, b);
        }
    }
}

// Direct form for what neither the block kernel nor the long-filter path takes (taps longer than the
// vector, where the reference uses the centre taps only, time_freq/mod.rs:284-288): one output per
// thread, taps streamed from L2, wrap-around by modular indexing.  O(N*M): a correctness net.
template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_conv_direct(const T* __restrict__ x, T* __restrict__ y,
                                                     const T* __restrict__ h, long long n,
                                                     long long count, long long conv_len,
                                                     long long total)
{
    long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total) return;
    long long vec = gi / n, i = gi % n;
    const T* xv = x + vec * n * (CPLX ? 2 : 1);
    long long pos = (i + conv_len) % n;
    if (CPLX) {
        T sr = 0, si = 0;
        for (long long k = 0; k < count; ++k) {
            pos = pos > 0 ? pos - 1 : n - 1;
            T ar = xv[2 * pos], ai = xv[2 * pos + 1], br = h[2 * k], bi = h[2 * k + 1];
            sr = sr + (ar * br - ai * bi);
            si = si + (ar * bi + ai * br);
        }
        y[2 * gi] = sr;
        y[2 * gi + 1] = si;
    } else {
        T sr = 0;
        for (long long k = 0; k < count; ++k) {
            pos = pos > 0 ? pos - 1 : n - 1;
            sr = sr + xv[pos] * h[k];
        }
        y[gi] = sr;
    }
}

template <typename T>
static size_t conv_lds_bytes() { return (size_t)(CONV_L + (CONV_L >> 4) + 16 * 17) * sizeof(cpx<T>); }

// Filter spectrum for the block kernel: hs[0..L) = FFT_L(zero-padded taps) in natural order,
// UNSCALED (the block kernel folds the 1/L of its unnormalised inverse transform into the load).  Exactly one of
// (taps_dev, h_freq_dev) is used: `taps` complex time-domain taps, or an UNSCALED length-L
// spectrum handed in by GpuSupport::overlap_discard.
template <typename T>
int conv_prepare_spectrum(const T* taps_dev, size_t taps, const T* h_freq_dev, T* hs, hipStream_t s)
{
    constexpr int L = CONV_L;
    if (h_freq_dev) {
        BDSP_HIP_TRY(hipMemcpyAsync(hs, h_freq_dev, sizeof(cpx<T>) * L, hipMemcpyDeviceToDevice, s));
        return BDSP_OK;
    }
    if (taps == 0 || taps > (size_t)L) return BDSP_ERR_ARG_LENGTH;
    // one launch: the taps are zero-padded while they are loaded (in_valid), plain I/O path
    FftIo<T> io{};
    io.n = L; io.in_stride = L; io.out_stride = L; io.flags = 0; io.window_id = -1;
    io.window_alpha = 0; io.in_scale = (T)1; io.in_valid = taps;
    io.in = taps_dev; io.out = hs;
    return fft_pow2<T>(io, nullptr, nullptr, 1, false, s);
}

// The fused overlap-save launch on a prepared spectrum.  in/out: `batch` contiguous complex
// vectors of `points`.  Block b reads x[(b*V + in_off + n) mod points] and writes outputs
// b*V + out_off + m, m < V, below `points`; all blocks (nblocks_limit = 0) or only the first
// nblocks_limit.  last_block_out (optional): the full L-point time-domain result of block
// nblocks_limit goes there.
template <typename T>
size_t conv_block_step(size_t points, size_t taps, bool real_data)
{
    if (conv_v2_applies(points, taps)) return conv_v2_block_step(taps);
    size_t V = (size_t)CONV_L - (taps - 1);
    if (real_data) { if (V >= 32) V &= ~(size_t)31; }
    else if (V >= 16) V &= ~(size_t)15;
    return V;
}
template size_t conv_block_step<float>(size_t, size_t, bool);
template size_t conv_block_step<double>(size_t, size_t, bool);

template <typename T>
int conv_run_blocks(const T* in, T* out, size_t points, size_t batch, const T* hs, size_t taps,
                    long long in_off, long long out_off, size_t nblocks_limit, T* last_block_out,
                    hipStream_t s, bool real_data, bool hs_is_taps)
{
    constexpr int L = CONV_L;
    // real data with real taps handed in as taps: the second-generation kernel on pairs of real blocks
    // (LAB, BDSP_CONV_REAL_PREP: also on a prepared spectrum, so that the workgroups need not transform the taps themselves)
    static const bool real_prep = lab_flag("BDSP_CONV_REAL_PREP");
    if (real_data && (hs_is_taps || real_prep) && !last_block_out && nblocks_limit == 0 && out_off == 0 && in_off == -(long long)(taps / 2) &&
        taps >= 1 && taps - 1 <= 3 * (size_t)L / 4 && conv_v2_applies(points, taps))
        return conv_v2_run<T>(in, out, points, batch, hs, taps, 0, 0, hs_is_taps, s, true);
    // complex data: the second-generation kernel (conv_v2.hip) whenever the call is a run of whole blocks
    if (!real_data && !last_block_out && taps >= 1 && taps - 1 <= 3 * (size_t)L / 4 && conv_v2_applies(points, taps)) {
        const long long V2 = (long long)conv_v2_block_step(taps);
        if (out_off >= 0 && out_off % V2 == 0 && in_off + (long long)(taps / 2) == out_off)
            return conv_v2_run<T>(in, out, points, batch, hs, taps, (size_t)(out_off / V2), nblocks_limit, hs_is_taps, s);
    }
    if (taps == 0 || taps - 1 > 3 * (size_t)L / 4 || points == 0) {
        set_last_error("convolve_overlap_save: taps out of range for the block kernel");
        return BDSP_ERR_UNSUPPORTED;
    }
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(L, &wtab));
    if (points >= (size_t(1) << 31) || batch > 65535) {
        set_last_error("convolve_overlap_save: vector too long or batch above 65535");
        return BDSP_ERR_UNSUPPORTED;
    }
    long long V = L - (long long)(taps - 1);
    if (real_data) { if (V >= 32) V &= ~31LL; }
    else if (V >= 16) V &= ~15LL; // must match the kernel's aligned block step
    long long per_vec = nblocks_limit ? (long long)nblocks_limit : ((long long)points + V - 1) / V;
    if (real_data) {
        if (nblocks_limit || last_block_out) { set_last_error("real block pairs: partial runs unsupported"); return BDSP_ERR_UNSUPPORTED; }
        per_vec = (per_vec + 1) / 2; // two real blocks per complex transform pair
    }
    size_t lds = conv_lds_bytes<T>();
    constexpr bool FAST = sizeof(T) == 4;
    auto kern = real_data ? k_overlap_save<T, FAST, true> : k_overlap_save<T, FAST, false>;
    if (lds > 64 * 1024)
        BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent-ish grid: enough workgroups to fill every CU at the occupancy LDS/VGPRs allow,
    // each walking blocks with a grid stride so the register-resident twiddles are loaded once
    int per_cu = sizeof(T) == 4 ? (BDSP_CONV_HL2 ? 4 : 3) : 2;
    long long want = (long long)num_cus() * per_cu;
    long long gx = (want + (long long)batch - 1) / (long long)batch;
    if (gx > per_vec) gx = per_vec;
    if (gx < 1) gx = 1;
    if (per_vec > 0) {
        hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)batch), dim3(256), lds, s,
                           static_cast<const void*>(in), static_cast<void*>(out),
                           reinterpret_cast<const cpx<T>*>(hs), wtab, (unsigned)points, (int)taps,
                           in_off, out_off, (unsigned)per_vec, (unsigned)points, hs_is_taps ? (real_data ? 6 : 2) : 0);
        BDSP_LAUNCH_CHECK();
    }
    if (last_block_out) {
        long long b = (long long)nblocks_limit;
        hipLaunchKernelGGL(kern, dim3(1, 1), dim3(256), lds, s,
                           static_cast<const void*>(in),
                           static_cast<void*>(last_block_out),
                           reinterpret_cast<const cpx<T>*>(hs), wtab, (unsigned)points, (int)taps,
                           in_off + b * V, 0LL, 1u, (unsigned)L, 1);
        BDSP_LAUNCH_CHECK();
    }
    return BDSP_OK;
}

template <typename T>
int convolve_overlap_save(const T* in, T* out, size_t points, size_t batch, const T* taps_dev,
                          size_t taps, long long in_off, long long out_off, size_t nblocks_limit,
                          T* last_block_out, const T* h_freq_dev, hipStream_t s)
{
    // with the taps in hand the block kernel transforms them itself (one launch for the whole convolution)
    static const bool no_fused_taps = lab_flag("BDSP_CONV_NO_FUSED_TAPS");
    if (!BDSP_CONV_HL2 && taps_dev && !h_freq_dev && !last_block_out && !no_fused_taps)
        return conv_run_blocks<T>(in, out, points, batch, taps_dev, taps, in_off, out_off, nblocks_limit, nullptr, s,
                                  false, true);
    WsBlock hsb;
    BDSP_TRY(hsb.alloc(sizeof(cpx<T>) * CONV_L, s));
    BDSP_TRY(conv_prepare_spectrum<T>(taps_dev, taps, h_freq_dev, hsb.as<T>(), s));
    return conv_run_blocks<T>(in, out, points, batch, hsb.as<T>(), taps, in_off, out_off,
                              nblocks_limit, last_block_out, s, false, false);
}

template <typename T>
int convolve_direct(const T* in, T* out, size_t points, size_t batch, const T* taps_dev,
                    size_t taps, bool is_complex, hipStream_t s)
{
    // (h', count, conv_len) as time_freq/mod.rs:284-296
    size_t start = 0, count = taps;
    long long conv_len = (long long)(taps - taps / 2);
    if (taps > points) {
        size_t center = taps / 2, cl = points / 2;
        start = center - cl;
        count = 2 * cl;
        conv_len = (long long)cl;
    }
    long long total = (long long)points * (long long)batch;
    if (total == 0) return BDSP_OK;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (is_complex)
        hipLaunchKernelGGL((k_conv_direct<T, true>), dim3(grid), dim3(256), 0, s, in, out,
                           taps_dev + 2 * start, (long long)points, (long long)count, conv_len, total);
    else
        hipLaunchKernelGGL((k_conv_direct<T, false>), dim3(grid), dim3(256), 0, s, in, out,
                           taps_dev + start, (long long)points, (long long)count, conv_len, total);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template int convolve_overlap_save<float>(const float*, float*, size_t, size_t, const float*, size_t,
                                          long long, long long, size_t, float*, const float*, hipStream_t);
template int convolve_overlap_save<double>(const double*, double*, size_t, size_t, const double*, size_t,
                                           long long, long long, size_t, double*, const double*, hipStream_t);
template int conv_prepare_spectrum<float>(const float*, size_t, const float*, float*, hipStream_t);
template int conv_prepare_spectrum<double>(const double*, size_t, const double*, double*, hipStream_t);
template int conv_run_blocks<float>(const float*, float*, size_t, size_t, const float*, size_t, long long, long long, size_t, float*, hipStream_t, bool, bool);
template int conv_run_blocks<double>(const double*, double*, size_t, size_t, const double*, size_t, long long, long long, size_t, double*, hipStream_t, bool, bool);
template int convolve_direct<float>(const float*, float*, size_t, size_t, const float*, size_t, bool, hipStream_t);
template int convolve_direct<double>(const double*, double*, size_t, size_t, const double*, size_t, bool, hipStream_t);

} // namespace bdsp

// ---- helpers for the generic (any power-of-two fft_len) GpuSupport::overlap_discard path --------
namespace bdsp {

// z[b][k] *= h[k] * scale   (multiply_vector of the OpenCL backend, ocl_kernels32.rs:68-78, with
// the 1/fft_len that clFFT's inverse applied folded in)
template <typename T>
__global__ __launch_bounds__(256) void k_mul_bcast(cpx<T>* __restrict__ z, const cpx<T>* __restrict__ h,
                                                    size_t l, size_t total, T scale)
{
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        cpx<T> hv = h[g % l];
        hv.x = hv.x * scale;
        hv.y = hv.y * scale;
        z[g] = cmul(z[g], hv);
    }
}

// x[b*step + dst_off + m] = z[b][skip + m], m < l - skip, for b < nb
template <typename T>
__global__ __launch_bounds__(256) void k_scatter_valid(const cpx<T>* __restrict__ z, cpx<T>* __restrict__ x,
                                                        size_t l, size_t skip, size_t step, size_t dst_off,
                                                        size_t nb, size_t x_points)
{
    size_t per = l - skip, total = per * nb;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        size_t b = g / per, m = g % per;
        size_t o = b * step + dst_off + m;
        if (o < x_points) x[o] = z[b * l + skip + m];
    }
}

template <typename T>
int mul_bcast(T* z, const T* h, size_t l, size_t nb, T scale, hipStream_t s)
{
    size_t total = l * nb;
    if (total == 0) return BDSP_OK;
    size_t blocks = (total + 255) / 256, cap = (size_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_mul_bcast<T>), dim3((unsigned)blocks), dim3(256), 0, s,
                       reinterpret_cast<cpx<T>*>(z), reinterpret_cast<const cpx<T>*>(h), l, total, scale);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template <typename T>
int scatter_valid(const T* z, T* x, size_t l, size_t skip, size_t step, size_t dst_off, size_t nb,
                  size_t x_points, hipStream_t s)
{
    size_t total = (l - skip) * nb;
    if (total == 0) return BDSP_OK;
    size_t blocks = (total + 255) / 256, cap = (size_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_scatter_valid<T>), dim3((unsigned)blocks), dim3(256), 0, s,
                       reinterpret_cast<const cpx<T>*>(z), reinterpret_cast<cpx<T>*>(x), l, skip, step,
                       dst_off, nb, x_points);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template int mul_bcast<float>(float*, const float*, size_t, size_t, float, hipStream_t);
template int mul_bcast<double>(double*, const double*, size_t, size_t, double, hipStream_t);
template int scatter_valid<float>(const float*, float*, size_t, size_t, size_t, size_t, size_t, size_t, hipStream_t);
template int scatter_valid<double>(const double*, double*, size_t, size_t, size_t, size_t, size_t, size_t, hipStream_t);

} // namespace bdsp
