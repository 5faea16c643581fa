// conv_v2_impl.h -- the fused overlap-save block kernel, second generation; included by conv_v2_f32.hip and conv_v2_f64.hip with
// BDSP_CONV_T set (round 6: one code object per precision -- a process's first f32 convolve_signal no longer loads the f64
// instantiations, 3.5 MB of code for the two together).
//
// Same mathematics as conv.hip's k_overlap_save (reference: overlap_discard, convolution.rs:304-461; result
// y[i] = sum_k x[(i + ceil(M/2) - 1 - k) mod N] h[k], time_freq/mod.rs:455-473): per 4096-point block
// load -> FFT -> x H -> IFFT -> store of the valid part, one launch.  What changed, each item measured on MI355X
// with tools/lab/conv_lab.hip (16M points x 1024 taps, random data, three rotating inputs; the first-generation
// kernel runs 74-76 us on the same boxes):
//   * NO MERGE POINTS IN THE BLOCK LOOP.  The first generation chose per block between plain and wrap-around loads
//     and predicated every store; hipcc's code around those joins cost 4.5 us (loads) and 3.9 us (stores) per launch
//     although the branches themselves are free.  Here the few blocks whose window wraps around the end of the
//     vector are taken first, by the first workgroups, through the general code; the loop over interior blocks has
//     plain contiguous loads and stores whole 256-point rows without a predicate.
//   * WHOLE-ROW STORES need the first valid output of a block on a row boundary: the taps are DELAYED by
//     d = 256 R0 - (M-1) samples (R0 = ceil((M-1)/256), a template parameter: a run-time R0 measured 7 us slower), so
//     z[256 R0 ...] are the block's outputs and V = 4096 - 256 R0.  Every store is then a full 128-byte line as well
//     (the first generation's started 8 bytes into one).
//   * A STATIC SKEW OF THE BLOCK COUNT.  A CU holds three workgroups; the hardware issues the OLDEST wave first, so
//     the workgroup dispatched first runs ~1.5x faster than the third and, with equal shares, finished 20 us before it
//     (per-workgroup timelines: 41 / 50 / 63 us), leaving the CU a third full.  Equalising the PROGRESS (a dynamic
//     ticket queue per XCD) measured slower (74-76 us): the CU is at its best with one workgroup running unimpeded and
//     the others filling its gaps.  So the shares follow the dispatch order instead: the first third of the grid
//     takes ~43 % of the blocks, the second ~37 %, the last ~20 % (whole rounds of G/3 blocks: 9 / 8 / 4.3 rounds at
//     16M points).  Only speed depends on the dispatch order.  62-63 us = 0.53-0.54 of the 8 TB/s roofline.
//   * exchange A in a layout without write conflicts (fft_core.h LAYOUT3: the PMC's 20 % LDS conflict cycles were all
//     on the first generation's 16-byte store pairs): 61.7 -> 61.3 us.
//   * measured and NOT adopted: decimation-in-frequency / -time transforms with a wave-private second exchange
//     (4 barriers per block instead of 8: 65-68 us with the skew, 2 workgroups per CU), all twiddles in registers at
//     2 per CU (64.4-66.8), register or LDS-DMA prefetch of the next block (72-77), 4 workgroups per CU (60.6 with a
//     four-way skew: no better than three).  Wavefront shuffles (tools/ubench/permlane_exchange.hip): trading two
//     register-index bits for lane bits 4 and 5 with v_permlane32_swap / v_permlane16_swap costs 28 ns per workgroup
//     exchange and CU against 217 ns for a full four-bit exchange through LDS, but only those two lane bits are
//     reachable that way (a select-and-shuffle exchange of two other bits measured 625 ns), so a transform built on
//     them needs radix 16 x 4 x 16 x 4 -- 55 more packed instructions per transform than 16 x 16 x 16 on the unit
//     that is already the busiest; not built.
#include "bdsp_internal.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace bdsp {

constexpr int L2 = 4096;

template <typename T>
struct ConvV2Args {
    const cpx<T>* x;
    cpx<T>* y;
    const cpx<T>* hs;   // taps (hs_is_taps) or the UNSCALED, undelayed L-point spectrum of the taps
    const cpx<T>* wtab; // exp(-2 pi i m / 4096)
    unsigned n;      // points per vector
    unsigned taps;
    unsigned b_first, b_end; // blocks of each vector to compute
    unsigned nb_lo, nb_hi;   // the interior ones among them: window inside [0, n), all V outputs below n
    unsigned batch;
    unsigned na, nbb;        // interior blocks (all vectors together) given to dispatch groups 0 and 1
    int hs_is_taps;
    unsigned groups;         // dispatch groups = workgroups per CU: 3 (f32), 2 (f64)
    unsigned stagger_us;     // LAB experiment (BDSP_CONV_STAGGER_US): dispatch groups after the first start this much later
};

// (ablations: a value the compiler must treat as defined / as used, without an instruction)
template <typename C> static __device__ __forceinline__ void abl_def(C& v) { asm volatile("" : "=v"(v.x), "=v"(v.y)); }
template <typename C> static __device__ __forceinline__ void abl_use(const C& v) { asm volatile("" : : "v"(v.x), "v"(v.y)); }

static __device__ __forceinline__ unsigned xcd_contiguous(unsigned bid, unsigned g)
{
    // workgroup w runs on XCD w % 8 (observed; only speed depends on it): give every XCD a contiguous run of the
    // blocks of a round, so that the M-1 input samples neighbouring blocks share are an L2 hit
    return (g & 7) == 0 ? (bid & 7) * (g >> 3) + (bid >> 3) : bid;
}

// f64 (round 2, late): the same structure with TWO workgroups per CU (74 KB of LDS and 240 registers each), two dispatch
// groups, the generic padded exchange layouts and the six-value split of the stage-3 twiddles (fifteen would not fit).
// REAL: the vector holds n REAL samples and the taps are real: the real blocks 2p and 2p+1 travel through the complex
// transform pair as real and imaginary part (convolution with a real filter is real-linear, so they come out
// separated).  "Block" then means such a PAIR; x, y and hs are read as arrays of T.
// ABL (LAB build only, BDSP_CONV_ABL=<bits>, R0 = 4): timing-only ablations of the interior loop -- 1 no global loads,
// 2 no global stores, 8 no transform (3 = arithmetic + exchanges alone, 8 = the memory skeleton alone); the output is
// garbage.  They put a measured bound next to the f64 and real-data kernels' roofline fractions.
// NTS: the interior blocks' results are STREAMED (non-temporal stores) -- for results the 256 MB Infinity Cache cannot hold
// anyway (round 5, conv_v2_streams_result below; round 3 measured the compile-time form on the headline step: +8 us, its
// 128 MB result is read back from the cache by the transform that follows).
template <typename T, int R0, bool BATCHED, bool REAL = false, int ABL = 0, bool NTS = false>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 3 : 2) void k_overlap_save_v2(ConvV2Args<T> a)
{
    constexpr int L = L2;
    constexpr unsigned V = L - 256 * R0, OV = 256 * R0;
    constexpr bool F32 = sizeof(T) == 4;
    using C32 = cpx<T>;
    using F = WgFft<T, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C32* lds = reinterpret_cast<C32*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const T hscale = (T)1 / (T)L;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    // Twiddles.  f32 (round 3): stages 2 and 3 are twiddled 16-point transforms in FMA form (fft_core.h dft16_tw: the
    // fifteen input twiddles ride on the multiply-adds of four radix-2 layers, 96 packed instructions instead of
    // 30 + 76) and need eight table values per stage and thread -- all sixteen in registers, no LDS table; the
    // inverse's last stage does not compute the R0 rows the block discards.  Lab (tools/lab/conv_lab.hip, k_v3):
    // 61.0 -> 59.8 us, 158 VGPRs.  f64 runs stage 2 in the same FMA form with its eight values read from a 2 KB LDS
    // table and stage 3 from FOUR held values {w^8, w^4, w^2, w} (the other four are products with constants, sixteen
    // multiply-adds per transform): 16 registers of twiddles instead of the 24 of the former six-value split
    // (w^r = w^(4a) w^b, nine extra complex multiplies per transform) -- the f64 kernel lives at its 256-register limit.
    constexpr int TW3_HELD = 2;
    C32 hreg[16], tw2f[F32 ? 8 : 1], tw3f[F32 ? 8 : 1], tw3q[TW3_HELD]; // (f64: {w^2, w})
    C32* tw2l = lds + F::LDS_ELEMS;
    const C32* tw2p = tw2l + (t & 15) * 9;
    if constexpr (F32) {
        F::template load_twiddles16_fma<16>(tw2f, t, tww);
        F::template load_twiddles16_fma<256>(tw3f, t, tww);
    } else {
        if (t < 128) {
            const int k = t >> 3, j = t & 7, e = 16 * k; // the j-th value of load_twiddles16_fma<16> for thread column k
            tw2l[k * 9 + j] = a.wtab[j == 0 ? 8 * e : j == 1 ? 4 * e : j == 2 ? 2 * e : j == 3 ? 2 * e + L / 8 : e + (j - 4) * (L / 16)];
        }
        if constexpr (TW3_HELD == 2) { tw3q[0] = a.wtab[2 * t]; tw3q[1] = a.wtab[t]; }
        else F::template load_twiddles16_fma4<256>(tw3q, t, tww);
        __syncthreads();
    }
    auto stage3 = [&](C32 (&v)[16], auto D, auto P) {
        constexpr int DIR = decltype(D)::value;
        if constexpr (F32) dft16_tw<DIR, decltype(P)::value>(&v[0], tw3f);
        else {
            C32 tl[8];
            expand_twiddles16_fma<TW3_HELD>(tw3q, tl);
            dft16_tw<DIR, decltype(P)::value>(&v[0], tl);
        }
    };
    auto stage2 = [&](C32 (&v)[16], auto D) {
        constexpr int DIR = decltype(D)::value;
        if constexpr (F32) dft16_tw<DIR>(&v[0], tw2f);
        else {
            C32 tl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tl[j] = tw2p[j];
            dft16_tw<DIR>(&v[0], tl);
        }
    };
    constexpr int PRUNE = R0 <= 8 ? R0 : 0;

    auto forward = [&](C32 (&v)[16]) {
        F::template compute<16, 1, -1>(v, t, tww);
        __syncthreads(); // the previous transform's last gather is done
        if constexpr (F32) F::scatter_a3(v, t, lds); else F::scatter_a(v, t, lds);
        __syncthreads();
        if constexpr (F32) F::gather_a3(v, t, lds); else F::gather_a(v, t, lds);
        stage2(v, std::integral_constant<int, -1>{});
        __syncthreads();
        if constexpr (F32) F::scatter_b3(v, t, lds); else F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        stage3(v, std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{});
    };
    auto inverse = [&](C32 (&v)[16]) {
        F::template compute<16, 1, 1>(v, t, tww);
        __syncthreads();
        if constexpr (F32) F::scatter_a3(v, t, lds); else F::scatter_a(v, t, lds);
        __syncthreads();
        if constexpr (F32) F::gather_a3(v, t, lds); else F::gather_a(v, t, lds);
        stage2(v, std::integral_constant<int, 1>{});
        __syncthreads();
        if constexpr (F32) F::scatter_b3(v, t, lds); else F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        stage3(v, std::integral_constant<int, 1>{}, std::integral_constant<int, PRUNE>{});
    };

    // ---- the filter spectrum, delayed by d samples, x 1/L, in register r of thread t: H'[t + 256 r]
    const unsigned d = OV - (a.taps - 1);
    if (a.hs_is_taps) {
        C32 hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned i = ut + 256u * r;
            if constexpr (REAL) hv[r] = (i >= d && i - d < a.taps) ? C32{reinterpret_cast<const T*>(a.hs)[i - d], (T)0} : C32{(T)0, (T)0};
            else hv[r] = (i >= d && i - d < a.taps) ? a.hs[i - d] : C32{(T)0, (T)0};
        }
        forward(hv);
#pragma unroll
        for (int r = 0; r < 16; ++r) hreg[r] = C32{hv[r].x * hscale, hv[r].y * hscale};
    } else {
        // a delay by d samples is the linear phase exp(-2 pi i k d / L) on bin k
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned k = ut + 256u * r;
            const C32 hv = a.hs[k];
            hreg[r] = cmul(C32{hv.x * hscale, hv.y * hscale}, a.wtab[(k * d) & (L - 1)]);
        }
    }
    auto transform = [&](C32 (&v)[16]) {
        forward(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
        inverse(v);
    };

    const long long in_off = -(long long)(a.taps / 2);
    const unsigned G = gridDim.x;
    // ---- blocks that wrap around the ends of their vector (or whose outputs run past it): general code, taken first
    {
        const unsigned nw = (a.nb_lo - a.b_first) + (a.b_end - a.nb_hi), total_w = nw * a.batch;
        for (unsigned w = blockIdx.x; w < total_w; w += G) {
            const unsigned vec = w / nw, k = w % nw;
            const unsigned b = k < a.nb_lo - a.b_first ? a.b_first + k : a.nb_hi + (k - (a.nb_lo - a.b_first));
            C32 v[16];
            if constexpr (REAL) {
                const T* xr = reinterpret_cast<const T*>(a.x) + (size_t)vec * a.n;
                T* yr = reinterpret_cast<T*>(a.y) + (size_t)vec * a.n;
                T part[2][16];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    long long sb = ((long long)(2 * b + half) * V + in_off) % (long long)a.n;
                    if (sb < 0) sb += a.n;
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[half][r] = xr[((unsigned long long)sb + ut + 256u * r) % a.n];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = C32{part[0][r], part[1][r]};
                transform(v);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const long long obase = (long long)(2 * b + half) * V - OV;
                    const long long room = (long long)a.n - obase;
                    const unsigned lim = room <= 0 ? 0u : (room > L ? (unsigned)L : (unsigned)room);
                    T* yb = yr + obase;
#pragma unroll
                    for (int r = R0; r < 16; ++r) {
                        const unsigned np = ut + 256u * r;
                        if (np < lim) yb[np] = half ? v[r].y : v[r].x;
                    }
                }
                continue;
            }
            const C32* xv = a.x + (size_t)vec * a.n;
            C32* yv = a.y + (size_t)vec * a.n;
            long long sb = ((long long)b * V + in_off) % (long long)a.n;
            if (sb < 0) sb += a.n;
            const unsigned idx = (unsigned)sb + ut;
            if (a.n >= (unsigned)L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned i = idx + 256u * r;
                    if (i >= a.n) i -= a.n;
                    v[r] = xv[i];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = xv[(idx + 256u * r) % a.n];
            }
            transform(v);
            // z[np], OV <= np < OV + V, is output b V + np - OV; only outputs below n exist
            const long long obase = (long long)b * V - OV;
            const long long room = (long long)a.n - obase;
            const unsigned lim = room <= 0 ? 0u : (room > L ? (unsigned)L : (unsigned)room);
            C32* yb = yv + obase;
#pragma unroll
            for (int r = R0; r < 16; ++r) {
                const unsigned np = ut + 256u * r;
                if (np < lim) yb[np] = v[r];
            }
        }
    }
    // ---- interior blocks: dispatch groups (one workgroup of each per CU) with skewed shares (see the header)
    const unsigned ni = a.nb_hi - a.nb_lo, total = ni * a.batch, gs = G / a.groups;
    const unsigned grp = blockIdx.x / gs;
    if (grp >= a.groups) return;
#ifdef BDSP_LAB
    if (a.stagger_us && grp > 0) { // (bounded: s_memrealtime counts 100 MHz ticks)
        const unsigned long long t0 = wall_clock64(), lim = (unsigned long long)a.stagger_us * 100ull * grp;
        while (wall_clock64() - t0 < lim) __builtin_amdgcn_s_sleep(32);
    }
#endif
    const unsigned lo = grp == 0 ? 0u : (grp == 1 ? a.na : a.na + a.nbb);
    const unsigned hi = grp == 0 ? a.na : ((grp == 1 && a.groups == 3) ? a.na + a.nbb : total);
    // (Round 3, measured and NOT adopted: RUNS of consecutive blocks per workgroup.  Block b + 1's first R0 rows are block
    // b's last R0 input rows -- the same registers of the same threads -- so a workgroup that walks consecutive blocks
    // can keep them: 12 instead of 16 loads per block at 1024 taps.  Lab kernel k_v5 (tools/lab/conv_lab.hip): 60.1 ->
    // 58.7 us on two boxes, 60.6 -> 60.2 on a third; in the library the kernel alone measured equal (60.3 vs 60.6 us) and
    // the step convolution -> FFT 7 us SLOWER (185.5 -> 192.5 us, three A/B rounds): with 768 runs the freshly written
    // lines are spread over the whole result when the kernel ends and their write-back lands in the transform's first
    // pass, whereas the strided sweep below leaves one compact window.)
    const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
    for (unsigned id = lo + w2; id < hi; id += gs) {
        unsigned vec = 0, b = a.nb_lo + id;
        if (BATCHED) { vec = id / ni; b = a.nb_lo + id % ni; }
        if constexpr (REAL) {
            const T* x0 = reinterpret_cast<const T*>(a.x) + ((size_t)vec * a.n + ((long long)(2 * b) * V + in_off));
            T* y0 = reinterpret_cast<T*>(a.y) + ((size_t)vec * a.n + ((long long)(2 * b) * V - OV));
            C32 v[16];
            if constexpr (ABL & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) abl_def(v[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = C32{x0[ut + 256u * r], x0[V + ut + 256u * r]};
            }
            if constexpr (!(ABL & 8)) transform(v);
            if constexpr (ABL & 2) {
#pragma unroll
                for (int r = R0; r < 16; ++r) abl_use(v[r]);
            } else {
#pragma unroll
                for (int r = R0; r < 16; ++r) {
                    y0[ut + 256u * r] = v[r].x;
                    y0[V + ut + 256u * r] = v[r].y;
                }
            }
            continue;
        }
        const C32* xb = a.x + ((size_t)vec * a.n + ((long long)b * V + in_off));
        C32* yb = a.y + ((size_t)vec * a.n + ((long long)b * V - OV));
        C32 v[16];
        if constexpr (ABL & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) abl_def(v[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if defined(BDSP_LAB) && defined(BDSP_CONV_NTL)
                v[r] = nt_load(&xb[ut + 256u * r]); // (A/B build, round 5: the input is dead once read)
#else
                v[r] = xb[ut + 256u * r];
#endif
            }
        }
        if constexpr (!(ABL & 8)) transform(v);
        if constexpr (ABL & 2) {
#pragma unroll
            for (int r = R0; r < 16; ++r) abl_use(v[r]);
            continue;
        }
#pragma unroll
        for (int r = R0; r < 16; ++r) {
            // (non-temporal stores measured 7 us slower on the FFT that follows: the result would leave the caches;
            // non-temporal loads of x made no difference)
            // (round 3, -DBDSP_CONV_NT in the lab build: streaming stores pay only when the result exceeds the cache --
            // 16M f64 points (256 MB) 129 -> 123 us, 64 x 1M f32 (512 MB) 232 -> 228 -- and cost the headline step 8 us)
            if constexpr (NTS) nt_store(&yb[ut + 256u * r], v[r]);
            else yb[ut + 256u * r] = v[r];
        }
    }
}

#ifdef BDSP_LAB
// ------------------------------------------------------------------------------------------------------------------
// LAB experiment, round 5 (BDSP_CONV_V3=1; complex f64, 770 ... 1025 taps: the block step of the product's R0 = 4): the third candidate of VERDICT r04's item 4 --
// the same block, 512 threads x EIGHT points per thread, 4096 = 8 x 8 x 8 x 8.  Half the registers per wave (v and H are
// 32 VGPRs each instead of 64: a 128-register budget), so a CU holds two 512-thread workgroups = FOUR waves per SIMD where
// the product kernel has two, at the price of a fourth stage and a third LDS exchange per transform (12 barriers per
// block instead of 8).  Rows are 512 points: R0 = ceil((M - 1) / 512) = 2 discarded rows for 1024 taps, V = 3072 as in the
// product.  Stage-2 / stage-3 twiddles (four values per thread: dft8_tw) from LDS tables of 8 x 4 and 64 x 4 entries,
// stage 4 from two held values {w^2, w} (w^4 by squaring, w W8 by a constant).  128 VGPRs, no scratch, four waves per SIMD.
// *Measured* (tools/conv_probe.py: correct on its first run, 6.3e-16 rel-L2 from the product kernel's result): 145.5-150.2 us
// against 120.5-123.1 -- twice the waves buy nothing, the fourth stage and the third exchange cost 20 %.  Not adopted.
template <int R0, bool NTS>
__global__ __launch_bounds__(512, 4) void k_overlap_save_v3(ConvV2Args<double> a)
{
    using T = double;
    constexpr int L = L2, NTH = 512;
    constexpr unsigned V = L - 512 * R0, OV = 512 * R0;
    using C64 = cpx<T>;
    using F = WgFft<T, L, NTH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C64* lds = reinterpret_cast<C64*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const T hscale = (T)1 / (T)L;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    C64* tw2l = lds + F::LDS_ELEMS;       // [8][4]
    C64* tw3l = tw2l + 8 * 4;             // [64][4]
    if (t < 32) {
        const int k = t >> 2, j = t & 3, e = k * (L / 64);
        tw2l[t] = a.wtab[j == 0 ? 4 * e : j == 1 ? 2 * e : j == 2 ? e : e + L / 8];
    }
    if (t < 256) {
        const int k = t >> 2, j = t & 3, e = k * (L / 512);
        tw3l[t] = a.wtab[j == 0 ? 4 * e : j == 1 ? 2 * e : j == 2 ? e : e + L / 8];
    }
    const C64 w4q[2] = {a.wtab[(2 * t) & (L - 1)], a.wtab[t]}; // stage 4 (NS = 512): e = t -> {w^2, w}
    __syncthreads();
    const C64* tw2p = tw2l + (t & 7) * 4;
    const C64* tw3p = tw3l + (t & 63) * 4;
    auto stage_tab = [&](C64 (&v)[8], const C64* tp, auto D) {
        const C64 tl[4] = {tp[0], tp[1], tp[2], tp[3]};
        dft8_tw<decltype(D)::value>(&v[0], tl);
    };
    auto stage4 = [&](C64 (&v)[8], auto D) {
        const T h = (T)0.70710678118654752440;
        const C64 w2 = w4q[0], w1 = w4q[1];
        const C64 tl[4] = {C64{(w2.x - w2.y) * (w2.x + w2.y), (T)2 * w2.x * w2.y}, w2, w1, C64{(w1.x + w1.y) * h, (w1.y - w1.x) * h}};
        dft8_tw<decltype(D)::value>(&v[0], tl);
    };
    auto xform = [&](C64 (&v)[8], auto D) {
        constexpr int DIR = decltype(D)::value;
        F::template compute<8, 1, DIR>(v, t, tww);
        __syncthreads(); // the previous transform's last gather is done
        F::template scatter<8, 1>(v, t, lds);
        __syncthreads();
        F::template gather<8>(v, t, lds);
        stage_tab(v, tw2p, D);
        __syncthreads();
        F::template scatter<8, 8>(v, t, lds);
        __syncthreads();
        F::template gather<8>(v, t, lds);
        stage_tab(v, tw3p, D);
        __syncthreads();
        F::template scatter<8, 64>(v, t, lds);
        __syncthreads();
        F::template gather<8>(v, t, lds);
        stage4(v, D);
    };
    // ---- the filter spectrum, delayed by d samples, x 1/L: H'[t + 512 r] in register r
    C64 hreg[8];
    const unsigned d = OV - (a.taps - 1);
    {
        C64 hv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned i = ut + 512u * r;
            hv[r] = (i >= d && i - d < a.taps) ? a.hs[i - d] : C64{(T)0, (T)0};
        }
        xform(hv, std::integral_constant<int, -1>{});
#pragma unroll
        for (int r = 0; r < 8; ++r) hreg[r] = C64{hv[r].x * hscale, hv[r].y * hscale};
    }
    auto transform = [&](C64 (&v)[8]) {
        xform(v, std::integral_constant<int, -1>{});
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = cmul(v[r], hreg[r]);
        xform(v, std::integral_constant<int, 1>{});
    };
    const long long in_off = -(long long)(a.taps / 2);
    const unsigned G = gridDim.x;
    // ---- blocks that wrap around the ends of the vector: general code, taken first
    {
        const unsigned nw = (a.nb_lo - a.b_first) + (a.b_end - a.nb_hi), total_w = nw * a.batch;
        for (unsigned w = blockIdx.x; w < total_w; w += G) {
            const unsigned vec = w / nw, k = w % nw;
            const unsigned b = k < a.nb_lo - a.b_first ? a.b_first + k : a.nb_hi + (k - (a.nb_lo - a.b_first));
            const C64* xv = a.x + (size_t)vec * a.n;
            C64* yv = a.y + (size_t)vec * a.n;
            long long sb = ((long long)b * V + in_off) % (long long)a.n;
            if (sb < 0) sb += a.n;
            C64 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = xv[((unsigned long long)sb + ut + 512u * r) % a.n];
            transform(v);
            const long long obase = (long long)b * V - OV;
            const long long room = (long long)a.n - obase;
            const unsigned lim = room <= 0 ? 0u : (room > L ? (unsigned)L : (unsigned)room);
            C64* yb = yv + obase;
#pragma unroll
            for (int r = R0; r < 8; ++r) {
                const unsigned np = ut + 512u * r;
                if (np < lim) yb[np] = v[r];
            }
        }
    }
    // ---- interior blocks: two dispatch groups with skewed shares, as in the product kernel
    const unsigned ni = a.nb_hi - a.nb_lo, total = ni * a.batch, gs = G / a.groups;
    const unsigned grp = blockIdx.x / gs;
    if (grp >= a.groups) return;
    const unsigned lo = grp == 0 ? 0u : a.na;
    const unsigned hi = grp == 0 ? a.na : total;
    const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
    for (unsigned id = lo + w2; id < hi; id += gs) {
        const unsigned vec = id / ni, b = a.nb_lo + id % ni;
        const C64* xb = a.x + ((size_t)vec * a.n + ((long long)b * V + in_off));
        C64* yb = a.y + ((size_t)vec * a.n + ((long long)b * V - OV));
        C64 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = xb[ut + 512u * r];
        transform(v);
#pragma unroll
        for (int r = R0; r < 8; ++r) {
            if constexpr (NTS) nt_store(&yb[ut + 512u * r], v[r]);
            else yb[ut + 512u * r] = v[r];
        }
    }
}
#endif // BDSP_LAB

template <typename T, int R0>
static int launch_v2(const ConvV2Args<T>& a, unsigned grid, size_t lds, hipStream_t s, bool real, bool nts)
{
#ifdef BDSP_LAB
    if constexpr (R0 == 4) {
        if (const char* e = lab_env("BDSP_CONV_ABL")) {
            const int abl = atoi(e);
#define BDSP_ABL(N)                                                                                                     \
    if (abl == N) {                                                                                                     \
        auto kern = real ? k_overlap_save_v2<T, R0, true, true, N> : k_overlap_save_v2<T, R0, true, false, N>;          \
        if (lds > 64 * 1024) BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);                                                     \
        BDSP_LAUNCH_CHECK();                                                                                            \
        return BDSP_OK;                                                                                                 \
    }
            BDSP_ABL(3) BDSP_ABL(8)
#undef BDSP_ABL
        }
    }
#endif
    // streamed results: complex data only, through the batched instantiation (it serves single vectors too)
    if (!real && nts) {
        auto kern = k_overlap_save_v2<T, R0, true, false, 0, true>;
        if (lds > 64 * 1024) BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
        BDSP_LAUNCH_CHECK();
        return BDSP_OK;
    }
    if (real) {
        auto kern = k_overlap_save_v2<T, R0, true, true>; // (the batched instantiation serves single vectors too)
        if (lds > 64 * 1024) BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    } else if (a.batch > 1) {
        auto kern = k_overlap_save_v2<T, R0, true>;
        if (lds > 64 * 1024) BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    } else {
        auto kern = k_overlap_save_v2<T, R0, false>;
        if (lds > 64 * 1024) BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// results above this size are streamed: see k_overlap_save_v2 NTS (set from the measurements of round 5)
// *Measured* (tools/conv_probe.py, profiles/r05_conv_probe.txt): 16M complex f64 points (256 MB in, 256 MB out), twelve interleaved
// runs each way in two processes: 121.5 / 123.0 -> 118.8 / 120.3 us (-2.2 %; single runs move by +-5 % on one box); 64 x 1M f32
// (512 MB) 225 -> 222 us for the kernel and 629 -> 632 for convolve -> fft, so f32 batches are left alone; the headline's 128 MB
// result must NOT be streamed (step 181 -> 192 us: the transform reads it from the cache).
template <typename T>
static bool conv_v2_streams_result(size_t points, size_t batch)
{
    return sizeof(T) == 8 && (double)points * (double)batch * 2.0 * sizeof(T) > 192.0 * 1024 * 1024;
}

// Per-CALL override of the dispatch-group shares (bdsp_hip_dev_convolve_ex sets it around its own launch): the
// calling thread's, so that no other thread's launches see it and the pair can never be read torn.
extern thread_local int t_conv_share_a, t_conv_share_b;
#ifdef BDSP_CONV_F32_TU // (the precision-independent pieces live in the f32 TU)
thread_local int t_conv_share_a = -1, t_conv_share_b = -1;
// block step of the second-generation kernel: V = 4096 - 256 ceil((M-1)/256)
size_t conv_v2_block_step(size_t taps)
{
    const size_t r0 = taps <= 1 ? 1 : (taps - 1 + 255) / 256;
    return (size_t)L2 - 256 * r0;
}

void conv_v2_set_shares(int first_pct, int second_pct)
{
    t_conv_share_a = first_pct;
    t_conv_share_b = second_pct;
}

bool conv_v2_applies(size_t points, size_t taps)
{
    static const bool off = lab_flag("BDSP_CONV_V1");
    return !off && taps >= 1 && taps - 1 <= 3 * (size_t)L2 / 4 && points >= 1 && points < (size_t(1) << 31);
}
#endif

// Blocks [first_block, first_block + nblocks) (nblocks = 0: all from first_block on) of every vector of the batch.
// real: `points` REAL samples per vector and real taps (hs_is_taps); blocks are then PAIRS of real blocks.
template <typename T>
int conv_v2_run(const T* in, T* out, size_t points, size_t batch, const T* hs, size_t taps,
                size_t first_block, size_t nblocks, bool hs_is_taps, hipStream_t s, bool real)
{
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(L2, &wtab));
    const unsigned r0 = taps <= 1 ? 1u : (unsigned)((taps - 1 + 255) / 256);
    const long long V = L2 - 256 * (long long)r0;
    const long long U = real ? 2 : 1; // real blocks per kernel block
    const long long nb_all = (((long long)points + V - 1) / V + U - 1) / U;
    long long b0 = (long long)first_block, b1 = nblocks ? b0 + (long long)nblocks : nb_all;
    if (b1 > nb_all) b1 = nb_all;
    if (b0 >= b1 || batch == 0) return BDSP_OK;
    const long long in_off = -(long long)(taps / 2);
    // interior blocks: window [bV + in_off, + L) inside [0, n) (then all V outputs are below n as well, see below)
    long long lo = b0, hi = b1;
    // (a pair is interior when both of its real blocks are: the first one's window start and the last one's end decide)
    while (lo < hi && lo * U * V + in_off < 0) ++lo;
    while (hi > lo && ((hi * U - 1) * V + in_off + L2 > (long long)points || (hi * U - 1) * V + V > (long long)points)) --hi;
    if ((unsigned long long)(b1 - b0) * batch >= (1ull << 32) || batch > 0xffffffffull) {
        set_last_error("convolve_overlap_save: too many blocks");
        return BDSP_ERR_UNSUPPORTED;
    }
    constexpr unsigned GROUPS_MAX = sizeof(T) == 4 ? 3 : 2; // what the kernel's register budget allows per CU
    // (LAB: fewer workgroups per CU for the real-data kernel, whose 16M-sample job is 3.6 pairs per workgroup at three)
    static const unsigned lab_groups = [] { const char* e = lab_env("BDSP_CONV_GROUPS"); return e ? (unsigned)atoi(e) : 0u; }();
    const unsigned GROUPS = (lab_groups >= 2 && lab_groups <= GROUPS_MAX) ? lab_groups : GROUPS_MAX;
    ConvV2Args<T> a{};
    a.groups = GROUPS;
    a.stagger_us = [] { const char* e = lab_env("BDSP_CONV_STAGGER_US"); return e ? (unsigned)atoi(e) : 0u; }();
    // A complex result larger than the Infinity Cache can hold until its reader comes is streamed past the caches
    // (DESIGN.md 4.3, round 5); LAB: BDSP_CONV_NTS=0 / 1 forces the choice
    bool nts = !real && conv_v2_streams_result<T>(points, batch);
    if (const char* e = lab_env("BDSP_CONV_NTS")) nts = !real && atoi(e) != 0;
    a.x = reinterpret_cast<const cpx<T>*>(in);
    a.y = reinterpret_cast<cpx<T>*>(out);
    a.hs = reinterpret_cast<const cpx<T>*>(hs);
    a.wtab = wtab;
    a.n = (unsigned)points;
    a.taps = (unsigned)taps;
    a.b_first = (unsigned)b0; a.b_end = (unsigned)b1;
    a.nb_lo = (unsigned)lo; a.nb_hi = (unsigned)hi;
    a.batch = (unsigned)batch;
    a.hs_is_taps = hs_is_taps ? 1 : 0;
    // grid: GROUPS workgroups per CU, a multiple of 8 * GROUPS so that every dispatch group is a multiple of 8
    const unsigned long long interior = (unsigned long long)(hi - lo) * batch;
    const unsigned long long wrap = (unsigned long long)((lo - b0) + (b1 - hi)) * batch;
    const unsigned Q = 8 * GROUPS;
    unsigned grid = (unsigned)num_cus() * GROUPS;
    grid -= grid % Q;
    if (grid < Q) grid = Q;
    if (interior + wrap < grid) { // a small problem: no more workgroups than blocks (each one transforms the taps first)
        grid = (unsigned)((interior + wrap + Q - 1) / Q * Q);
        if (grid < Q) grid = Q;
    }
    const unsigned gs = grid / GROUPS;
    // shares in whole rounds of gs blocks.  Three groups: ~43 % / ~37 % / rest (measured optimum 9 / 8 / 4.3 rounds of
    // 21.3); two groups: ~55 % / rest (12 of 21.3 rounds measured best for two workgroups per CU)
    const unsigned long long rounds = (interior + gs - 1) / gs;
    // (bdsp_hip_dev_convolve_ex overrides the percentages for its own call: the guard test times equal shares against these)
    const int sa = t_conv_share_a, sb = t_conv_share_b;
    const unsigned long long pa = sa > 0 ? (unsigned long long)sa : (GROUPS == 3 ? 43 : 55);
    const unsigned long long pb = GROUPS == 3 ? (sa > 0 ? (unsigned long long)sb : 37) : 0;
    unsigned long long ra = (rounds * pa + 50) / 100, rb = (rounds * pb + 50) / 100;
    if (rounds && ra == 0) ra = 1;
    unsigned long long na = ra * gs, nbb = rb * gs;
    if (na > interior) na = interior;
    if (na + nbb > interior) nbb = interior - na;
    a.na = (unsigned)na;
    a.nbb = (unsigned)nbb;
    const size_t lds = (size_t)(sizeof(T) == 4 ? WgFft<T, L2, 256>::LDS_ELEMS3 : WgFft<T, L2, 256>::LDS_ELEMS + 16 * 17) * sizeof(cpx<T>);
#ifdef BDSP_LAB
    if constexpr (sizeof(T) == 8) {
        if (lab_flag("BDSP_CONV_V3") && !real && r0 == 4 && GROUPS == 2) {
            using F3 = WgFft<double, L2, 512>;
            const size_t lds3 = (size_t)(F3::LDS_ELEMS + 8 * 4 + 64 * 4) * sizeof(cpx<double>);
            auto kern = nts ? k_overlap_save_v3<2, true> : k_overlap_save_v3<2, false>;
            BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds3, s, a);
            BDSP_LAUNCH_CHECK();
            return BDSP_OK;
        }
    }
#endif
    switch (r0) {
#define BDSP_R0(N) case N: return launch_v2<T, N>(a, grid, lds, s, real, nts);
        BDSP_R0(1) BDSP_R0(2) BDSP_R0(3) BDSP_R0(4) BDSP_R0(5) BDSP_R0(6)
        BDSP_R0(7) BDSP_R0(8) BDSP_R0(9) BDSP_R0(10) BDSP_R0(11) BDSP_R0(12)
#undef BDSP_R0
    default: break;
    }
    set_last_error("convolve_overlap_save: taps out of range for the block kernel");
    return BDSP_ERR_UNSUPPORTED;
}

template int conv_v2_run<BDSP_CONV_T>(const BDSP_CONV_T*, BDSP_CONV_T*, size_t, size_t, const BDSP_CONV_T*, size_t, size_t, size_t, bool, hipStream_t, bool);

} // namespace bdsp
