// conv_lab -- a standalone bench + checker for variants of the fused overlap-save block kernel (complex f32,
// 16M points, 1024 taps by default).  Every variant is checked against a double-precision direct convolution at
// a few hundred output positions (both wrap-around ends, block seams, random interior) and timed with HIP
// events on random data after a clock warm-up, rotating three input buffers.
//
//   make -C tools/lab && tools/lab/conv_lab [points] [taps]
//
// Variants (see the kernels below):
//   base     the shipped kernel's structure: Stockham 16x16x16, 8 barriers per block, split stage-3 twiddles,
//            stage-2 twiddles in an LDS table, predicated stores
//   base+as  the same with ALIGNED stores: the taps are delayed by round_up(M-1,16)-(M-1) samples so that the valid
//            outputs of a block start at a multiple of 16 points; whole rows are stored without a predicate
//   dif      decimation-in-frequency forward / decimation-in-time inverse: no autosort, the spectrum stays in
//            digit-reversed order (H is stored to match), the exchange between the second and third stage only
//            moves data between lanes of ONE wavefront (no barrier), all twiddles in registers; 4 barriers per block,
//            2 workgroups per CU
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <type_traits>
#include <vector>
#include "fft_core.h"
using namespace bdsp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int L = 4096;
using C = cpx<float>;

struct Args {
    const C* x; C* y; const C* hs; const C* wtab;
    unsigned n; int ov;  // z[ov] is the first valid output of a block (>= taps-1; the taps are delayed to match)
    unsigned V;          // block step: block b yields outputs [bV, bV+V) from z[ov .. ov+V)
    long long in_off;    // block b reads x[(b*V + in_off + i) mod n]
    unsigned blocks;
    unsigned* q;             // dynamic block queues: 8 counters 128 bytes apart + a done counter, zero between launches
    unsigned long long* clk; // [4]: s_memtime and s_memrealtime at the start and end of workgroup 0
    const C* wtab8 = nullptr; // exp(-2 pi i m / 8192), k_v6 only
};

// ------------------------------------------------------------------------------------------- shared pieces
__device__ __forceinline__ void load_block(const Args& a, unsigned b, unsigned V, unsigned ut, C (&d)[16])
{
    long long base = (long long)b * V + a.in_off;
    if (base >= 0 && base + L <= (long long)a.n) {
        const C* xb = a.x + base;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = xb[ut + 256u * r];
    } else {
        long long sb = base % (long long)a.n;
        if (sb < 0) sb += a.n;
        const unsigned idx = (unsigned)sb + ut;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned i = idx + 256u * r;
            if (i >= a.n) i -= a.n;
            d[r] = a.x[i];
        }
    }
}

template <bool ALIGNED>
__device__ __forceinline__ void store_block(const Args& a, unsigned b, unsigned V, unsigned ut, const C (&v)[16])
{
    const unsigned ov = (unsigned)a.ov;
    const long long obase = (long long)b * V - ov; // output index of z[0]
    if (ALIGNED && (ov & 255u) == 0 && (long long)b * V + V <= (long long)a.n) {
        // whole rows, no predicate: rows r0 .. 15 are the valid ones (uniform)
        C* yb = a.y + obase;
        const unsigned r0 = ov >> 8;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((unsigned)r >= r0) yb[ut + 256u * r] = v[r];
        return;
    }
    long long room = (long long)a.n - obase;
    unsigned lim = room <= 0 ? 0u : (room > L ? (unsigned)L : (unsigned)room);
    if (lim > ov + V) lim = ov + V;
    C* yb = a.y + obase;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned np = ut + 256u * r;
        if (np >= ov && np < lim) yb[np] = v[r];
    }
}

__device__ __forceinline__ unsigned xcd_contiguous(unsigned bid, unsigned g)
{
    return (g & 7) == 0 ? (bid & 7) * (g >> 3) + (bid >> 3) : bid;
}

// ------------------------------------------------------------------------------------------- base
template <bool ALIGNED>
__global__ __launch_bounds__(256, 3) void k_base(Args a)
{
    using F = WgFft<float, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* lds = reinterpret_cast<C*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
#ifdef LAB_PROBE
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (threadIdx.x == 0 && a.clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    const unsigned V = a.V;
    const float hscale = 1.0f / L;
    auto tw = [&](int mm) { return a.wtab[mm]; };
    C tw3a[3], tw3b[3], hreg[16];
    C* tw2l = lds + F::LDS_ELEMS;
    F::template load_twiddles16_split<256>(tw3a, tw3b, t, tw);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        C hv = a.hs[ut + 256u * r];
        hreg[r] = C{hv.x * hscale, hv.y * hscale};
    }
    if (t < 240) {
        int k = t / 15, r = t % 15 + 1;
        tw2l[k * 17 + r - 1] = a.wtab[r * k * 16];
    }
    __syncthreads();
    const C* tw2p = tw2l + (t & 15) * 17;
    const unsigned G = gridDim.x, wl = xcd_contiguous(blockIdx.x, G);
    for (unsigned b = wl; b < a.blocks; b += G) {
        C v[16];
        load_block(a, b, V, ut, v);
        F::template compute<16, 1, -1>(v, t, tw);
        __syncthreads();
        F::scatter_a(v, t, lds);
        __syncthreads();
        F::gather_a(v, t, lds);
        F::template compute_pre<16, 16, -1>(v, tw2p);
        __syncthreads();
        F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        F::template compute_pre16_split<256, -1>(v, tw3a, tw3b);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
        F::template compute<16, 1, 1>(v, t, tw);
        __syncthreads();
        F::scatter_a(v, t, lds);
        __syncthreads();
        F::gather_a(v, t, lds);
        F::template compute_pre<16, 16, 1>(v, tw2p);
        __syncthreads();
        F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        F::template compute_pre16_split<256, 1>(v, tw3a, tw3b);
        store_block<ALIGNED>(a, b, V, ut, v);
    }
#ifdef LAB_PROBE
    if (threadIdx.x == 0 && a.clk) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* q = a.clk + 4 * blockIdx.x;
        q[0] = clk_r0;
        q[1] = __builtin_amdgcn_s_memrealtime();
        q[2] = __builtin_amdgcn_s_memtime() - clk_t0;
        q[3] = xcc & 7;
    }
#endif
}

// ------------------------------------------------------------------------------------------- dif
// Index bits of the block-local time index i = 256 i2 + 16 i1 + i0 and of the frequency k = k0 + 16 k1 + 256 k2.
//   load      thread t = 16 i1 + i0, register i2
//   DFT16 over i2 -> k0;  x w4096^((16 i1 + i0) k0)              (15 per-thread twiddles, registers)
//   CROSS exchange: register <-> thread bits 4..7:  thread 16 k0 + i0, register i1         (barriers)
//   DFT16 over i1 -> k1;  x w256^(i0 k1)                          (15 per-thread twiddles, registers)
//   WAVE-PRIVATE exchange: register <-> lane bits 0..3: thread 16 k0 + k1, register i0     (no barrier)
//   DFT16 over i0 -> k2:  thread 16 k0 + k1 holds X[k0 + 16 k1 + 256 r] in register r
// and the inverse runs the same steps backwards with conjugated twiddles.
// LDS layouts (elements of 8 bytes; all conflict-free for ds_write_b64 16-lane groups / ds_read_b64 32-lane groups):
//   cross:   A(i1, i0, k0) = i0 + 16 k0 + 272 i1            (one pad row of 16 per 256)
//   private: P(g, a, r)    = 272 g + 17 r + a  per wave, g = lane >> 4, a = lane & 15, region of 1088 elements
constexpr int CROSS_ELEMS = 16 * 272, PRIV_ELEMS = 4 * 272;

template <int DIR>
__device__ __forceinline__ void tw_apply(C (&v)[16], const C (&tw)[15])
{
#pragma unroll
    for (int r = 1; r < 16; ++r) v[r] = twmul<DIR>(v[r], tw[r - 1]);
}

template <bool ALIGNED, int WPC>
__global__ __launch_bounds__(256, WPC) void k_dif(Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* cross = reinterpret_cast<C*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const int lane = t & 63, wave = t >> 6;
    C* priv = cross + CROSS_ELEMS + wave * PRIV_ELEMS;
    const unsigned V = a.V;
    const float hscale = 1.0f / L;
    // twiddles: tw1[k0-1] = w4096^(t k0), tw2[k1-1] = w256^((t & 15) k1)
    C tw1[15], tw2[15], hreg[16];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        tw1[r - 1] = a.wtab[(t * r) & (L - 1)];
        tw2[r - 1] = a.wtab[(16 * (t & 15) * r) & (L - 1)];
    }
    {
        // after the forward transform thread t'' = 16 k0 + k1 holds X[k0 + 16 k1 + 256 r]
        const unsigned k0 = ut >> 4, k1 = ut & 15;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            C hv = a.hs[k0 + 16u * k1 + 256u * r];
            hreg[r] = C{hv.x * hscale, hv.y * hscale};
        }
    }
    // exchange addressing, base(thread) + constant(register)
    C* const cw = cross + (t & 15) + 272 * (t >> 4);       // writer (i1 = t>>4, i0 = t&15), + 16 r   (r = k0)
    const C* const cr = cross + (t & 15) + 16 * (t >> 4);  // reader (k0 = t>>4, i0 = t&15), + 272 r  (r = i1)
    C* const pw = priv + 272 * (lane >> 4) + (lane & 15);        // + 17 r
    const C* const pr = priv + 272 * (lane >> 4) + 17 * (lane & 15); // + r
    const unsigned G = gridDim.x, wl = xcd_contiguous(blockIdx.x, G);
    for (unsigned b = wl; b < a.blocks; b += G) {
        C v[16];
        load_block(a, b, V, ut, v);
        // ---------------- forward, decimation in frequency
        dft16<-1>(v);
        tw_apply<-1>(v, tw1);
        __syncthreads(); // the previous block's inverse cross gather is done
#pragma unroll
        for (int r = 0; r < 16; ++r) cw[16 * r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cr[272 * r];
        dft16<-1>(v);
        tw_apply<-1>(v, tw2);
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[17 * r] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = pr[r];
        dft16<-1>(v);
        // ---------------- spectrum product
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
        // ---------------- inverse, decimation in time
        dft16<1>(v);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[17 * r] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = pr[r];
        tw_apply<1>(v, tw2);
        dft16<1>(v);
        __syncthreads(); // everybody has finished the forward cross gather
        // inverse cross: writer is thread (k0 = t>>4, i0), register i1 -> same cell A(i1, i0, k0)
#pragma unroll
        for (int r = 0; r < 16; ++r) const_cast<C*>(cr)[272 * r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cw[16 * r];
        tw_apply<1>(v, tw1);
        dft16<1>(v);
        store_block<ALIGNED>(a, b, V, ut, v);
    }
}


// ------------------------------------------------------------------------------------------- ub
// The structure of tools/ubench/mem_interference.hip's "shipped shape" (NOT a correct convolution: every stage
// multiplies by 15 register twiddles, loads are clamped instead of wrapped, rows 4..15 are stored unconditionally).
// MODE bit 0: real wrap-around load_block; bit 1: real predicated store_block; bit 2: stage-2 twiddles from the LDS
// table; bit 3: split stage-3 twiddles; bit 4: no twiddles in the first stage
template <int MODE>
__global__ __launch_bounds__(256, 3) void k_ub(Args a)
{
    using F = WgFft<float, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* lds = reinterpret_cast<C*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const unsigned V = a.V;
    const float hscale = 1.0f / L;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    C tw[15], hreg[16], tw3a[3], tw3b[3];
#pragma unroll
    for (int r = 0; r < 15; ++r) tw[r] = a.wtab[(t * (r + 1)) & (L - 1)];
    F::template load_twiddles16_split<256>(tw3a, tw3b, t, tww);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        C hv = a.hs[ut + 256u * r];
        hreg[r] = C{hv.x * hscale, hv.y * hscale};
    }
    C* tw2l = lds + F::LDS_ELEMS;
    if (t < 240) {
        int k = t / 15, r = t % 15 + 1;
        tw2l[k * 17 + r - 1] = a.wtab[r * k * 16];
    }
    __syncthreads();
    const C* tw2p = tw2l + (t & 15) * 17;
    const unsigned G = gridDim.x, wl = xcd_contiguous(blockIdx.x, G);
    for (unsigned b = wl; b < a.blocks; b += G) {
        C v[16];
        if (MODE & 1) load_block(a, b, V, ut, v);
        else {
            long long base = (long long)b * V + a.in_off;
            if (base < 0) base = 0;
            if (base + L > (long long)a.n) base = (long long)a.n - L;
            const C* xb = a.x + base;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = xb[ut + 256u * r];
        }
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            auto s1 = [&](auto D) { if (MODE & 16) F::template compute<16, 1, decltype(D)::value>(v, t, tww); else F::template compute_pre<16, 256, decltype(D)::value>(v, tw); };
            auto s2 = [&](auto D) { if (MODE & 4) F::template compute_pre<16, 16, decltype(D)::value>(v, tw2p); else F::template compute_pre<16, 256, decltype(D)::value>(v, tw); };
            auto s3 = [&](auto D) { if (MODE & 8) F::template compute_pre16_split<256, decltype(D)::value>(v, tw3a, tw3b); else F::template compute_pre<16, 256, decltype(D)::value>(v, tw); };
            auto run = [&](auto D) {
                s1(D);
                __syncthreads();
                F::scatter_a(v, t, lds);
                __syncthreads();
                F::gather_a(v, t, lds);
                s2(D);
                __syncthreads();
                F::scatter_b(v, t, lds);
                __syncthreads();
                F::gather_b(v, t, lds);
                s3(D);
            };
            if (dir == 0) {
                run(std::integral_constant<int, -1>{});
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
            } else run(std::integral_constant<int, 1>{});
        }
        if (MODE & 2) store_block<false>(a, b, V, ut, v);
        else {
            long long ob = (long long)b * V - a.ov;
            if (ob + L > (long long)a.n) ob = (long long)a.n - L;
            if (ob < 0) ob = 0;
            C* yb = a.y + ob;
#pragma unroll
            for (int r = 4; r < 16; ++r) yb[ut + 256u * r] = v[r];
        }
    }
}


// ------------------------------------------------------------------------------------------- v2
// One block loop without merge points: interior blocks only (plain contiguous loads, rows R0..15 stored whole, no
// predicate; the taps are delayed so that z[256 R0] is a block's first valid output); the few blocks whose window
// wraps around the end of the vector are taken FIRST, by the first workgroups, through the general load/store code.
//   CORE 0: Stockham 16x16x16 (8 barriers), CORE 1: DIF/DIT with the wave-private exchange (4 barriers)
//   TWR bit 0: stage-2 twiddles in registers (else LDS table), bit 1: all 15 stage-3 twiddles in registers (else split)
//   DBUF: two register sets take turns, so the next block's loads never wait for this block's stores to retire
template <int CORE, int R0, int TWR, int DBUF, int WPC>
__global__ __launch_bounds__(256, WPC) void k_v2(Args a, unsigned nb_lo, unsigned nb_hi)
{
    using F = WgFft<float, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* lds = reinterpret_cast<C*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const int lane = t & 63, wave = t >> 6;
#ifdef LAB_PROBE
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (threadIdx.x == 0 && a.clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    const unsigned V = a.V;
    const float hscale = 1.0f / L;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    C hreg[16];
    // Stockham twiddles
    C tw2r[(CORE != 1 && (TWR & 1)) ? 15 : 1], tw3r[(CORE != 1 && (TWR & 2)) ? 15 : 1], tw3a[3], tw3b[3];
    C* tw2l = lds + (CORE == 2 ? F::LDS_ELEMS3 : F::LDS_ELEMS);
    const C* tw2p = tw2l + (t & 15) * 17;
    // CORE 3 (round 3): L3 exchange layouts, stages 2 and 3 as twiddled 16-point transforms in FMA form (dft16_tw:
    // 96 packed instructions instead of 30 + 76, eight twiddle values per stage, all in registers); TWR bit 0: the
    // inverse's last stage does not compute the R0 rows the block discards
    C tw2f[CORE == 3 ? 8 : 1], tw3f[CORE == 3 ? 8 : 1];
    // DIF twiddles
    C tw1[CORE == 1 ? 15 : 1], tw2[CORE == 1 ? 15 : 1];
    C* const cross = lds;
    C* const priv = cross + CROSS_ELEMS + wave * PRIV_ELEMS;
    C* const cw = cross + (t & 15) + 272 * (t >> 4);
    C* const cr = cross + (t & 15) + 16 * (t >> 4);
    C* const pw = priv + 272 * (lane >> 4) + (lane & 15);
    const C* const pr = priv + 272 * (lane >> 4) + 17 * (lane & 15);
    if constexpr (CORE == 3) {
        F::template load_twiddles16_fma<16>(tw2f, t, tww);
        F::template load_twiddles16_fma<256>(tw3f, t, tww);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            C hv = a.hs[ut + 256u * r];
            hreg[r] = C{hv.x * hscale, hv.y * hscale};
        }
    } else if constexpr (CORE == 0 || CORE == 2) {
        if constexpr ((TWR & 1) != 0) {
#pragma unroll
            for (int r = 1; r < 16; ++r) tw2r[r - 1] = a.wtab[16 * r * (t & 15)];
        } else {
            if (t < 240) {
                int k = t / 15, r = t % 15 + 1;
                tw2l[k * 17 + r - 1] = a.wtab[r * k * 16];
            }
            __syncthreads();
        }
        if constexpr (TWR & 2) {
#pragma unroll
            for (int r = 1; r < 16; ++r) tw3r[r - 1] = a.wtab[(r * t) & (L - 1)];
        } else F::template load_twiddles16_split<256>(tw3a, tw3b, t, tww);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            C hv = a.hs[ut + 256u * r];
            hreg[r] = C{hv.x * hscale, hv.y * hscale};
        }
    } else {
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            tw1[r - 1] = a.wtab[(t * r) & (L - 1)];
            tw2[r - 1] = a.wtab[(16 * (t & 15) * r) & (L - 1)];
        }
        const unsigned k0 = ut >> 4, k1 = ut & 15;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            C hv = a.hs[k0 + 16u * k1 + 256u * r];
            hreg[r] = C{hv.x * hscale, hv.y * hscale};
        }
    }
    auto forward = [&](C (&v)[16]) {
        if constexpr (CORE == 3) {
            F::template compute<16, 1, -1>(v, t, tww);
            __syncthreads();
            F::scatter_a3(v, t, lds);
            __syncthreads();
            F::gather_a3(v, t, lds);
            dft16_tw<-1>(&v[0], tw2f);
            __syncthreads();
            F::scatter_b3(v, t, lds);
            __syncthreads();
            F::gather_b(v, t, lds);
            dft16_tw<-1>(&v[0], tw3f);
        } else if constexpr (CORE == 2) {
            constexpr int PX = (TWR & 4) ? 2 : ((TWR & 8) ? 0 : -1), PC = (TWR & 4) ? 0 : ((TWR & 8) ? 2 : -1);
            if constexpr (PC >= 0) __builtin_amdgcn_s_setprio(PC);
            F::template compute<16, 1, -1>(v, t, tww);
            if constexpr (PX >= 0) __builtin_amdgcn_s_setprio(PX);
            __syncthreads();
            F::scatter_a3(v, t, lds);
            __syncthreads();
            F::gather_a3(v, t, lds);
            if constexpr (PC >= 0) __builtin_amdgcn_s_setprio(PC);
            F::template compute_pre<16, 16, -1>(v, tw2p);
            if constexpr (PX >= 0) __builtin_amdgcn_s_setprio(PX);
            __syncthreads();
            F::scatter_b3(v, t, lds);
            __syncthreads();
            F::gather_b(v, t, lds);
            if constexpr (PC >= 0) __builtin_amdgcn_s_setprio(PC);
            if constexpr (TWR & 2) F::template compute_pre<16, 256, -1>(v, tw3r); else F::template compute_pre16_split<256, -1>(v, tw3a, tw3b);
        } else if constexpr (CORE == 0) {
            F::template compute<16, 1, -1>(v, t, tww);
            __syncthreads();
            F::scatter_a(v, t, lds);
            __syncthreads();
            F::gather_a(v, t, lds);
            if constexpr (TWR & 1) F::template compute_pre<16, 16, -1>(v, tw2r); else F::template compute_pre<16, 16, -1>(v, tw2p);
            __syncthreads();
            F::scatter_b(v, t, lds);
            __syncthreads();
            F::gather_b(v, t, lds);
            if constexpr (TWR & 2) F::template compute_pre<16, 256, -1>(v, tw3r); else F::template compute_pre16_split<256, -1>(v, tw3a, tw3b);
        } else {
            dft16<-1>(v);
            tw_apply<-1>(v, tw1);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) cw[16 * r] = v[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cr[272 * r];
            dft16<-1>(v);
            tw_apply<-1>(v, tw2);
#pragma unroll
            for (int r = 0; r < 16; ++r) pw[17 * r] = v[r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = pr[r];
            dft16<-1>(v);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
    };
    auto inverse = [&](C (&v)[16]) {
        if constexpr (CORE == 3) {
            F::template compute<16, 1, 1>(v, t, tww);
            __syncthreads();
            F::scatter_a3(v, t, lds);
            __syncthreads();
            F::gather_a3(v, t, lds);
            dft16_tw<1>(&v[0], tw2f);
            __syncthreads();
            F::scatter_b3(v, t, lds);
            __syncthreads();
            F::gather_b(v, t, lds);
            dft16_tw<1, ((TWR & 1) && R0 <= 8) ? R0 : 0>(&v[0], tw3f);
        } else if constexpr (CORE == 2) {
            constexpr int PX = (TWR & 4) ? 2 : ((TWR & 8) ? 0 : -1), PC = (TWR & 4) ? 0 : ((TWR & 8) ? 2 : -1);
            if constexpr (PC >= 0) __builtin_amdgcn_s_setprio(PC);
            F::template compute<16, 1, 1>(v, t, tww);
            if constexpr (PX >= 0) __builtin_amdgcn_s_setprio(PX);
            __syncthreads();
            F::scatter_a3(v, t, lds);
            __syncthreads();
            F::gather_a3(v, t, lds);
            if constexpr (PC >= 0) __builtin_amdgcn_s_setprio(PC);
            F::template compute_pre<16, 16, 1>(v, tw2p);
            if constexpr (PX >= 0) __builtin_amdgcn_s_setprio(PX);
            __syncthreads();
            F::scatter_b3(v, t, lds);
            __syncthreads();
            F::gather_b(v, t, lds);
            if constexpr (PC >= 0) __builtin_amdgcn_s_setprio(PC);
            if constexpr (TWR & 2) F::template compute_pre<16, 256, 1>(v, tw3r); else F::template compute_pre16_split<256, 1>(v, tw3a, tw3b);
        } else if constexpr (CORE == 0) {
            F::template compute<16, 1, 1>(v, t, tww);
            __syncthreads();
            F::scatter_a(v, t, lds);
            __syncthreads();
            F::gather_a(v, t, lds);
            if constexpr (TWR & 1) F::template compute_pre<16, 16, 1>(v, tw2r); else F::template compute_pre<16, 16, 1>(v, tw2p);
            __syncthreads();
            F::scatter_b(v, t, lds);
            __syncthreads();
            F::gather_b(v, t, lds);
            if constexpr (TWR & 2) F::template compute_pre<16, 256, 1>(v, tw3r); else F::template compute_pre16_split<256, 1>(v, tw3a, tw3b);
        } else {
            dft16<1>(v);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) pw[17 * r] = v[r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = pr[r];
            tw_apply<1>(v, tw2);
            dft16<1>(v);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) cr[272 * r] = v[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cw[16 * r];
            tw_apply<1>(v, tw1);
            dft16<1>(v);
        }
    };
    auto transform = [&](C (&v)[16]) { forward(v); inverse(v); };
    const unsigned G = gridDim.x, wl = xcd_contiguous(blockIdx.x, G);
    // the blocks that wrap around: [0, nb_lo) and [nb_hi, blocks), one per leading workgroup, general code
    {
        const unsigned nwrap = nb_lo + (a.blocks - nb_hi);
        for (unsigned w = blockIdx.x; w < nwrap; w += G) {
            const unsigned b = w < nb_lo ? w : nb_hi + (w - nb_lo);
            C v[16];
            load_block(a, b, V, ut, v);
            transform(v);
            store_block<false>(a, b, V, ut, v);
        }
    }
    auto load_fast = [&](unsigned b, C (&v)[16]) {
        const C* xb = a.x + ((long long)b * V + a.in_off);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = xb[ut + 256u * r];
    };
    auto store_fast = [&](unsigned b, const C (&v)[16]) {
        if constexpr (R0 > 0) {
            C* yb = a.y + ((long long)b * V - 256 * R0);
#pragma unroll
            for (int r = R0; r < 16; ++r) yb[ut + 256u * r] = v[r];
        } else {
            const int r0 = a.ov >> 8; // uniform
            C* yb = a.y + ((long long)b * V - a.ov);
#pragma unroll
            for (int r = 1; r < 16; ++r)
                if (r >= r0) yb[ut + 256u * r] = v[r];
        }
    };
    if constexpr (DBUF == 2 || DBUF == 3) {
        // Dynamic distribution: the interior blocks [nb_lo, nb_hi) are split into eight contiguous queues (one per XCD,
        // so that neighbouring blocks -- which share M-1 input samples -- are in flight on the same L2); a workgroup
        // takes its first block statically and every later one with an atomic ticket from its XCD's queue (then from
        // the other queues once its own is empty).  The ticket for the NEXT block is requested while this one is
        // being transformed and handed to the other waves through LDS.  Why: with a static grid-stride walk the
        // three workgroups of a CU finish 20 us apart (the oldest wave wins every issue arbitration), and the CU
        // idles through the tail.  The last workgroup to leave resets the counters for the next launch.
        __shared__ unsigned s_next[2];
        const unsigned nq = DBUF == 2 ? 8u : 1u;
        const unsigned total = nb_hi - nb_lo, per = (total + nq - 1) / nq;
        unsigned myq = (DBUF == 2) ? (blockIdx.x & 7u) : 0u;
        const unsigned first_local = DBUF == 2 ? (blockIdx.x >> 3) : blockIdx.x, prefill = DBUF == 2 ? (G >> 3) : G;
        auto q_lo = [&](unsigned q) { return nb_lo + q * per; };
        auto q_cnt = [&](unsigned q) { unsigned lo = q * per; return lo >= total ? 0u : (total - lo < per ? total - lo : per); };
        unsigned tried = 0; // thread 0: queues found empty so far (own queue first, then the neighbours)
        auto fetch = [&]() -> unsigned {
            while (tried < nq) {
                const unsigned q = (myq + tried) % nq;
                const unsigned v = atomicAdd(&a.q[q * 32], 1u) + (tried == 0 ? prefill : (DBUF == 2 ? (G >> 3) : G));
                if (v < q_cnt(q)) return q_lo(q) + v;
                ++tried;
            }
            return 0xffffffffu;
        };
        unsigned b = first_local < q_cnt(myq) ? q_lo(myq) + first_local : 0xffffffffu;
        if (b == 0xffffffffu) { // (only when a queue is shorter than the resident workgroups of its XCD)
            if (t == 0) s_next[1] = fetch();
            __syncthreads();
            b = s_next[1];
            __syncthreads();
        }
        unsigned it = 0;
        if constexpr (DBUF == 2) {
            // the ticket is requested by an UNTRACKED atomic (inline asm: the compiler must not wait for it at the
            // top of the block, where it would expose the whole round trip); it is awaited at the end of the block
            while (b != 0xffffffffu) {
                C v[16];
                load_fast(b, v);
                unsigned ticket = 0;
                if (t == 0 && tried == 0) {
                    unsigned* qp = &a.q[myq * 32];
                    const unsigned one = 1;
                    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(qp), "v"(one) : "memory");
                }
                transform(v);
                store_fast(b, v);
                if (t == 0) {
                    unsigned nb = 0xffffffffu;
                    if (tried == 0) {
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket)::"memory");
                        const unsigned vv = ticket + prefill;
                        if (vv < q_cnt(myq)) nb = q_lo(myq) + vv; else { tried = 1; nb = fetch(); }
                    } else nb = fetch();
                    s_next[it & 1] = nb;
                }
                __syncthreads();
                b = s_next[it & 1];
                ++it;
            }
        } else {
        while (b != 0xffffffffu) {
            C v[16];
            load_fast(b, v);
            if (t == 0) s_next[it & 1] = fetch();
            transform(v);
            store_fast(b, v);
            b = s_next[it & 1]; // written before the transform's barriers, read after them
            ++it;
        }
        }
        if (t == 0) {
            const unsigned old = atomicAdd(&a.q[8 * 32], 1u);
            if (old == G - 1) {
                for (unsigned q = 0; q < 8; ++q) a.q[q * 32] = 0;
                a.q[8 * 32] = 0;
            }
        }
    } else if constexpr (DBUF >= 3000000) {
        // HYBRID: whole static rounds per dispatch group (DBUF = 3000000 + 10000*RA + 100*RB + RC) and the blocks left
        // over handed out one by one from an atomic counter to whichever workgroup has finished its static share
        constexpr int RA = (DBUF - 3000000) / 10000, RB = (DBUF - 3000000) / 100 % 100, RC = (DBUF - 3000000) % 100;
        __shared__ unsigned s_tk;
        const unsigned total = nb_hi - nb_lo, gs = G / 3;
        unsigned n0 = RA * gs, n1 = RB * gs, n2 = RC * gs;
        if (n0 > total) n0 = total;
        if (n0 + n1 > total) n1 = total - n0;
        if (n0 + n1 + n2 > total) n2 = total - n0 - n1;
        const unsigned stat = n0 + n1 + n2;
        const unsigned grp = blockIdx.x / gs;
        const unsigned starts[4] = {0, n0, n0 + n1, stat};
        if (grp < 3) {
            const unsigned lo = nb_lo + starts[grp], hi = nb_lo + starts[grp + 1];
            const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
            for (unsigned b = lo + w2; b < hi; b += gs) {
                C v[16];
                load_fast(b, v);
                transform(v);
                store_fast(b, v);
            }
        }
        for (;;) {
            if (t == 0) s_tk = atomicAdd(&a.q[0], 1u);
            __syncthreads();
            const unsigned b = nb_lo + stat + s_tk;
            __syncthreads();
            if (b >= nb_hi) break;
            C v[16];
            load_fast(b, v);
            transform(v);
            store_fast(b, v);
        }
        if (t == 0) {
            const unsigned old = atomicAdd(&a.q[8 * 32], 1u);
            if (old == G - 1) { a.q[0] = 0; a.q[8 * 32] = 0; }
        }
    } else if constexpr (DBUF >= 1000000) {
        // four dispatch groups (4 workgroups per CU): DBUF = 1000000 + 10000*RA + 100*RB + RC rounds, the last group the rest
        constexpr int RA = (DBUF - 1000000) / 10000, RB = (DBUF - 1000000) / 100 % 100, RC = (DBUF - 1000000) % 100;
        const unsigned total = nb_hi - nb_lo, gs = G / 4;
        unsigned n0 = RA * gs, n1 = RB * gs, n2 = RC * gs;
        if (n0 > total) n0 = total;
        if (n0 + n1 > total) n1 = total - n0;
        if (n0 + n1 + n2 > total) n2 = total - n0 - n1;
        const unsigned grp = blockIdx.x / gs;
        const unsigned starts[5] = {0, n0, n0 + n1, n0 + n1 + n2, total};
        if (grp < 4) {
            const unsigned lo = nb_lo + starts[grp], hi = nb_lo + starts[grp + 1];
            const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
            for (unsigned b = lo + w2; b < hi; b += gs) {
                C v[16];
                load_fast(b, v);
                transform(v);
                store_fast(b, v);
            }
        }
    } else if constexpr (DBUF >= 2000) {
        // skew in whole ROUNDS: group g (of WPC groups of G/WPC workgroups, in dispatch order) takes RA, RB rounds
        // of G/WPC blocks each; the last group takes what is left.  DBUF = 2000 + 100*RA + RB (RB = 0 with two groups)
        constexpr int RA = (DBUF - 2000) / 100, RB = (DBUF - 2000) % 100;
        const unsigned total = nb_hi - nb_lo, gs = G / WPC;
        unsigned na = RA * gs, nbb = WPC == 3 ? RB * gs : 0;
        if (na > total) na = total;
        if (na + nbb > total) nbb = total - na;
        const unsigned grp = blockIdx.x / gs;
        const unsigned lo = grp == 0 ? nb_lo : ((grp == 1 && WPC == 3) ? nb_lo + na : nb_lo + na + nbb);
        const unsigned hi = grp == 0 ? nb_lo + na : ((grp == 1 && WPC == 3) ? nb_lo + na + nbb : nb_hi);
        const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
        if (grp < (unsigned)WPC)
            for (unsigned b = lo + w2; b < hi; b += gs) {
                C v[16];
                load_fast(b, v);
                transform(v);
                store_fast(b, v);
            }
    } else if constexpr (DBUF >= 1000) {
        // three-way static skew for 3 workgroups per CU: DBUF = 1000 + 100*a + b: shares a/100.. of group 0 and group 1
        constexpr int SA = (DBUF - 1000) / 100, SB = (DBUF - 1000) % 100;
        const unsigned total = nb_hi - nb_lo, third = G / 3;
        const unsigned na = (unsigned)((unsigned long long)total * SA / 100), nbb = (unsigned)((unsigned long long)total * SB / 100);
        const unsigned grp = blockIdx.x / third;
        const unsigned lo = grp == 0 ? nb_lo : (grp == 1 ? nb_lo + na : nb_lo + na + nbb);
        const unsigned hi = grp == 0 ? nb_lo + na : (grp == 1 ? nb_lo + na + nbb : nb_hi);
        const unsigned w2 = xcd_contiguous(blockIdx.x - grp * third, third);
        if (grp < 3)
            for (unsigned b = lo + w2; b < hi; b += third) {
                C v[16];
                load_fast(b, v);
                transform(v);
                store_fast(b, v);
            }
    } else if constexpr (DBUF >= 50) {
        // static SKEW: the workgroups dispatched first (blockIdx < G/2: the older of the two on their CU, which win the
        // issue arbitration and run ~1.5x faster) take a larger share of the blocks
        const unsigned total = nb_hi - nb_lo, half = G >> 1;
        const unsigned na = (unsigned)((unsigned long long)total * DBUF / 100);
        const bool first = blockIdx.x < half;
        const unsigned lo = first ? nb_lo : nb_lo + na, hi = first ? nb_lo + na : nb_hi;
        const unsigned w2 = xcd_contiguous(first ? blockIdx.x : blockIdx.x - half, half);
        for (unsigned b = lo + w2; b < hi; b += half) {
            C v[16];
            load_fast(b, v);
            transform(v);
            store_fast(b, v);
        }
    } else if constexpr (!DBUF) {
        for (unsigned b = nb_lo + wl; b < nb_hi; b += G) {
            C v[16];
            load_fast(b, v);
            transform(v);
            store_fast(b, v);
        }
    } else {
        // two register sets: while one block is transformed the next one's loads are in flight; a set is reloaded
        // half a block after its stores were issued, when they have retired
        unsigned b = nb_lo + wl;
        C va[16], vb[16];
        if (b < nb_hi) load_fast(b, va);
        for (; b < nb_hi; b += 2 * G) {
            const bool has_b = b + G < nb_hi;
            if (has_b) load_fast(b + G, vb);
            transform(va);
            store_fast(b, va);
            if (has_b) {
                forward(vb);
                if (b + 2 * G < nb_hi) load_fast(b + 2 * G, va);
                inverse(vb);
                store_fast(b + G, vb);
            }
        }
    }
#ifdef LAB_PROBE
    if (threadIdx.x == 0 && a.clk) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* q = a.clk + 4 * blockIdx.x;
        q[0] = clk_r0;
        q[1] = __builtin_amdgcn_s_memrealtime();
        q[2] = __builtin_amdgcn_s_memtime() - clk_t0;
        q[3] = xcc & 7;
    }
#endif
}




// Timing-only stand-in for a Linzer-Feig radix-4 form of dft16_tw: a twiddled radix-4 butterfly
// needs 11 packed instructions instead of the 12 of two radix-2 layers (t0 = a + w2 c [2], t1 = 2a - t0 [1],
// B = b + tau1 (i b) [1], u^ = B + W d [2], v^ = 2B - u^ [1], y0/y2 = t0 +- c1 u^ [2], y1/y3 = t1 -+ i c1 v^ [2]) --
// eight groups per 16-point transform, 88 instead of 96.  This copy of dft16_tw simply DROPS one "2a - s" per group
// (the result is garbage): what the block kernel would gain from the instruction count alone, before the 15 instead of
// 8 twiddle registers per stage the real form needs.
template <int DIR, int PRUNE = 0, typename CT>
__device__ __forceinline__ void dft16_tw_lfcount(CT* v, const CT* T)
{
    auto bf = [&](CT& a, CT& b, CT tw, auto ROT, bool drop) {
        const CT s = bf_tw_s<DIR, decltype(ROT)::value>(a, b, tw);
        if (!drop) b = bf_2a_minus_s(a, s);
        a = s;
    };
    using TR = std::integral_constant<bool, true>;
    using FA = std::integral_constant<bool, false>;
#pragma unroll
    for (int r = 0; r < 8; ++r) bf(v[r], v[r + 8], T[0], FA{}, r < 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) bf(v[r], v[r + 4], T[1], FA{}, r < 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) bf(v[8 + r], v[8 + r + 4], T[1], TR{}, r < 1);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        bf(v[r], v[r + 2], T[2], FA{}, r < 1);
        bf(v[4 + r], v[4 + r + 2], T[2], TR{}, r < 1);
        bf(v[8 + r], v[8 + r + 2], T[3], FA{}, false);
        bf(v[12 + r], v[12 + r + 2], T[3], TR{}, false);
    }
    // last layer as in dft16_tw (pruning included), two of its eight butterflies without their second output
    bf_tw<DIR, false, (0 >= PRUNE)>(v[0], v[1], T[4]);
    bf_tw<DIR, true, (4 >= PRUNE)>(v[2], v[3], T[4]);
    bf_tw<DIR, false, (2 >= PRUNE)>(v[4], v[5], T[6]);
    bf_tw<DIR, true, (6 >= PRUNE)>(v[6], v[7], T[6]);
    bf_tw<DIR, false, (1 >= PRUNE)>(v[8], v[9], T[5]);
    bf_tw<DIR, true, (5 >= PRUNE)>(v[10], v[11], T[5]);
    bf(v[12], v[13], T[7], FA{}, true);
    bf(v[14], v[15], T[7], TR{}, true);
}

// ------------------------------------------------------------------------------------------- v3 (round 3)
// The round-3 block kernel on its own: L3 exchange layouts, stages 2 and 3 as FMA-form twiddled 16-point transforms
// (dft16_tw), the inverse's last stage pruned by the R0 rows the block discards, WPC dispatch groups with the shares
// RA / RB (whole rounds; the last group takes the rest).  ABL switches pieces OFF for timing-only ablations (the
// output is then garbage): 1 global loads, 2 global stores, 4 LDS scatter/gather, 8 butterflies + filter multiply,
// 16 barriers.
// PF 1: two register sets take turns -- the next block's loads are issued half a block ahead (in the middle of the
// other set's transform); PF 2: the same with all sixteen twiddle values read from LDS tables (17 KB) instead of held in
// registers, which pays for the second register set at three workgroups per CU
template <int R0, int ABL, int WPC, int RA, int RB, int PF = 0>
__global__ __launch_bounds__(256, WPC) void k_v3(Args a, unsigned nb_lo, unsigned nb_hi)
{
    using F = WgFft<float, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* lds = reinterpret_cast<C*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const unsigned V = a.V;
    const float hscale = 1.0f / L;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    C hreg[16], tw2f[8], tw3f[8];
    F::template load_twiddles16_fma<16>(tw2f, t, tww);
    F::template load_twiddles16_fma<256>(tw3f, t, tww);
    C* const tab3 = lds + F::LDS_ELEMS3 + t;            // [j][256]
    C* const tab2 = lds + F::LDS_ELEMS3 + 2048 + (t & 15); // [j][16]
    if constexpr (PF == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            tab3[256 * j] = tw3f[j];
            if (t < 16) tab2[16 * j] = tw2f[j];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        C hv = a.hs[ut + 256u * r];
        hreg[r] = C{hv.x * hscale, hv.y * hscale};
    }
    __syncthreads();
    auto bar = [&]() { if constexpr (!(ABL & 16)) __syncthreads(); };
    auto half = [&](C (&v)[16], auto D) {
        constexpr int DIR = decltype(D)::value;
        if constexpr (!(ABL & 8)) F::template compute<16, 1, DIR>(v, t, tww);
        bar();
        if constexpr (!(ABL & 4)) F::scatter_a3(v, t, lds);
        bar();
        if constexpr (!(ABL & 4)) F::gather_a3(v, t, lds);
        if constexpr (!(ABL & 8)) {
            if constexpr (PF == 2) {
                C tl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tl[j] = tab2[16 * j];
                dft16_tw<DIR>(&v[0], tl);
            } else if constexpr (ABL & 256) dft16_tw_lfcount<DIR>(&v[0], tw2f);
            else dft16_tw<DIR>(&v[0], tw2f);
        }
        bar();
        if constexpr (!(ABL & 4)) F::scatter_b3(v, t, lds);
        bar();
        if constexpr (!(ABL & 4)) F::gather_b(v, t, lds);
        if constexpr (!(ABL & 8)) {
            if constexpr (PF == 2) {
                C tl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tl[j] = tab3[256 * j];
                dft16_tw<DIR, (DIR > 0 && R0 <= 8) ? R0 : 0>(&v[0], tl);
            } else if constexpr (ABL & 256) dft16_tw_lfcount<DIR, (DIR > 0 && R0 <= 8) ? R0 : 0>(&v[0], tw3f);
            else dft16_tw<DIR, (DIR > 0 && R0 <= 8) ? R0 : 0>(&v[0], tw3f);
        }
    };
    auto transform = [&](C (&v)[16]) {
        half(v, std::integral_constant<int, -1>{});
        if constexpr (!(ABL & 8)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
        }
        half(v, std::integral_constant<int, 1>{});
    };
    const unsigned G = gridDim.x;
    {
        // ABL bit 128: the wrap-around blocks go to the LAST workgroups of the grid (the last dispatch group, whose
        // highest-numbered workgroups have one block fewer than the others) instead of the first ones
        const unsigned nwrap = nb_lo + (a.blocks - nb_hi);
        const unsigned me = (ABL & 128) ? G - 1 - blockIdx.x : blockIdx.x;
        for (unsigned w = me; w < nwrap; w += G) {
            const unsigned b = w < nb_lo ? w : nb_hi + (w - nb_lo);
            C v[16];
            load_block(a, b, V, ut, v);
            transform(v);
            store_block<false>(a, b, V, ut, v);
        }
    }
    const unsigned total = nb_hi - nb_lo, gs = G / WPC;
    // RA / RB: whole rounds of gs blocks, or (values >= 64) sixteenths of a round -- shares need not be whole rounds: the
    // first (share mod gs) workgroups of a group then take one block more than the others
    unsigned na = RA >= 64 ? RA * gs / 16 : RA * gs, nbb = WPC >= 3 ? (RB >= 64 ? RB * gs / 16 : RB * gs) : 0;
    if (na > total) na = total;
    if (na + nbb > total) nbb = total - na;
    const unsigned grp = blockIdx.x / gs;
    if (grp >= (unsigned)WPC) return;
    const unsigned lo = grp == 0 ? nb_lo : ((grp == 1 && WPC == 3) ? nb_lo + na : nb_lo + na + nbb);
    const unsigned hi = grp == 0 ? nb_lo + na : ((grp == 1 && WPC == 3) ? nb_lo + na + nbb : nb_hi);
    const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
    if constexpr (PF != 0) {
        auto load_fast = [&](unsigned b, C (&v)[16]) {
            const C* xb = a.x + ((long long)b * V + a.in_off);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = xb[ut + 256u * r];
        };
        auto store_fast = [&](unsigned b, const C (&v)[16]) {
            C* yb = a.y + ((long long)b * V - 256 * R0);
#pragma unroll
            for (int r = R0; r < 16; ++r) yb[ut + 256u * r] = v[r];
        };
        unsigned b = lo + w2;
        C va[16], vb[16];
        if (b < hi) load_fast(b, va);
        for (; b < hi; b += 2 * gs) {
            const bool has_b = b + gs < hi;
            if (has_b) load_fast(b + gs, vb);
            transform(va);
            store_fast(b, va);
            if (has_b) {
                half(vb, std::integral_constant<int, -1>{});
#pragma unroll
                for (int r = 0; r < 16; ++r) vb[r] = cmul(vb[r], hreg[r]);
                if (b + 2 * gs < hi) load_fast(b + 2 * gs, va);
                half(vb, std::integral_constant<int, 1>{});
                store_fast(b + gs, vb);
            }
        }
        return;
    }
    for (unsigned b = lo + w2; b < hi; b += gs) {
        C v[16];
        if constexpr (ABL & 64) { // timing only: the same 32 KB as eight 16-byte loads per thread
            const float4* xb = reinterpret_cast<const float4*>(a.x + ((long long)b * V + a.in_off));
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 q = xb[ut + 256u * r];
                v[2 * r] = C{q.x, q.y};
                v[2 * r + 1] = C{q.z, q.w};
            }
        } else if constexpr (!(ABL & 1)) {
            const C* xb = a.x + ((long long)b * V + a.in_off);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = xb[ut + 256u * r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "=v"(v[r]));
        }
        transform(v);
        if constexpr (ABL & 32) { // timing only: the same 24 KB as six 16-byte stores per thread
            float4* yb = reinterpret_cast<float4*>(a.y + (long long)b * V);
#pragma unroll
            for (int r = R0; r < 16; r += 2) yb[ut + 128u * (r - R0)] = float4{v[r].x, v[r].y, v[r + 1].x, v[r + 1].y};
        } else if constexpr (!(ABL & 2)) {
            C* yb = a.y + ((long long)b * V - 256 * R0);
#pragma unroll
            for (int r = R0; r < 16; ++r) yb[ut + 256u * r] = v[r];
        } else {
#pragma unroll
            for (int r = R0; r < 16; ++r) asm volatile("" : : "v"(v[r]));
        }
    }
}


// ------------------------------------------------------------------------------------------- v4 (round 3): prefetch by hand
// Ablations of k_v3 (profiles/r03_conv_lab_variants.txt): the memory traffic alone takes 40 us at ONE workgroup per
// CU (32 KB in flight per CU saturate the fabric), butterflies + exchanges alone 37 us at three per CU, the kernel
// 58-61: a workgroup has requests in flight only during the ~2 us load phase of its ~8 us block, so the memory
// system idles a third of the time.  hipcc turns a C++ double buffer into load + immediate vmcnt(0).  Here the loads
// of block i+1 are issued by UNTRACKED inline-asm loads right after block i's data has been taken out of the
// prefetch registers, and awaited by an explicit s_waitcnt at the end of block i (vmcnt counts loads and stores in
// order: the twelve stores of block i, issued after the loads, may stay in flight).
template <int R0, int WPC, int RA, int RB, int MODE = 0>
__global__ __launch_bounds__(256, WPC) void k_v4(Args a, unsigned nb_lo, unsigned nb_hi)
{
    using F = WgFft<float, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* lds = reinterpret_cast<C*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const unsigned V = a.V;
    const float hscale = 1.0f / L;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    C hreg[16], tw2f[8], tw3f[8];
    F::template load_twiddles16_fma<16>(tw2f, t, tww);
    F::template load_twiddles16_fma<256>(tw3f, t, tww);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        C hv = a.hs[ut + 256u * r];
        hreg[r] = C{hv.x * hscale, hv.y * hscale};
    }
    auto half = [&](C (&v)[16], auto D) {
        constexpr int DIR = decltype(D)::value;
        F::template compute<16, 1, DIR>(v, t, tww);
        __syncthreads();
        F::scatter_a3(v, t, lds);
        __syncthreads();
        F::gather_a3(v, t, lds);
        dft16_tw<DIR>(&v[0], tw2f);
        __syncthreads();
        F::scatter_b3(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        dft16_tw<DIR, (DIR > 0 && R0 <= 8) ? R0 : 0>(&v[0], tw3f);
    };
    auto transform = [&](C (&v)[16]) {
        half(v, std::integral_constant<int, -1>{});
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
        half(v, std::integral_constant<int, 1>{});
    };
    const unsigned G = gridDim.x;
    {
        const unsigned nwrap = nb_lo + (a.blocks - nb_hi);
        for (unsigned w = blockIdx.x; w < nwrap; w += G) {
            const unsigned b = w < nb_lo ? w : nb_hi + (w - nb_lo);
            C v[16];
            load_block(a, b, V, ut, v);
            transform(v);
            store_block<false>(a, b, V, ut, v);
        }
    }
    const unsigned total = nb_hi - nb_lo, gs = G / WPC;
    unsigned na = RA * gs, nbb = WPC >= 3 ? RB * gs : 0;
    if (na > total) na = total;
    if (na + nbb > total) nbb = total - na;
    const unsigned grp = blockIdx.x / gs;
    if (grp >= (unsigned)WPC) return;
    const unsigned lo = grp == 0 ? nb_lo : ((grp == 1 && WPC == 3) ? nb_lo + na : nb_lo + na + nbb);
    const unsigned hi = grp == 0 ? nb_lo + na : ((grp == 1 && WPC == 3) ? nb_lo + na + nbb : nb_hi);
    const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
    const unsigned voff = ut * 8u;
    auto issue_loads = [&](unsigned b, C (&p)[16]) {
        const C* xb = a.x + ((long long)b * V + a.in_off);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const C* base = xb + 256 * r;
            asm volatile("global_load_dwordx2 %0, %2, %3\n\tglobal_load_dwordx2 %1, %2, %3 offset:2048"
                         : "=&v"(p[r]), "=&v"(p[r + 1]) : "v"(voff), "s"(base));
        }
    };
    auto wait_loads = [&](C (&p)[16]) {
        // all sixteen loads have landed; the (younger) twelve stores of the block may still be in flight
        asm volatile("s_waitcnt vmcnt(12)"
                     : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),
                       "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]));
    };
    unsigned b = lo + w2;
    if (b >= hi) return;
    C p[16];
    issue_loads(b, p);
    wait_loads(p);
    for (; b < hi; b += gs) {
        C v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = p[r];
        if (b + gs < hi) issue_loads(b + gs, p);
        transform(v);
        C* yb = a.y + ((long long)b * V - 256 * R0);
#pragma unroll
        for (int r = R0; r < 16; ++r) yb[ut + 256u * r] = v[r];
        wait_loads(p);
    }
}


// ------------------------------------------------------------------------------------------- v5 (round 3): consecutive blocks
// A workgroup takes a RUN of consecutive blocks instead of every gs-th one: block b + 1 starts V = 4096 - 256 R0 samples
// after block b, so its first R0 rows ARE block b's last R0 input rows -- the same registers of the same threads.  They
// are kept (2 R0 VGPRs) instead of loaded again: 12 instead of 16 loads per block at 1024 taps, a quarter less L2 -> CU
// traffic.  (The overlap used to be an L2 hit of a neighbouring workgroup's load; now only a run's FIRST block reads
// its overlap from memory.)  Shares as before: group g's workgroups get RA / RB / the rest "rounds" = blocks per run.
// ABL as in k_v3 (1 loads off, 2 stores off, 28 = memory only).
template <int R0, int ABL, int WPC, int RA, int RB>
__global__ __launch_bounds__(256, WPC) void k_v5(Args a, unsigned nb_lo, unsigned nb_hi)
{
    using F = WgFft<float, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* lds = reinterpret_cast<C*>(smem_raw);
    const int t = threadIdx.x;
    const unsigned ut = t;
    const unsigned V = a.V;
    const float hscale = 1.0f / L;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    C hreg[16], tw2f[8], tw3f[8];
    F::template load_twiddles16_fma<16>(tw2f, t, tww);
    F::template load_twiddles16_fma<256>(tw3f, t, tww);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        C hv = a.hs[ut + 256u * r];
        hreg[r] = C{hv.x * hscale, hv.y * hscale};
    }
    auto bar = [&]() { if constexpr (!(ABL & 16)) __syncthreads(); };
    auto half = [&](C (&v)[16], auto D) {
        constexpr int DIR = decltype(D)::value;
        if constexpr (!(ABL & 8)) F::template compute<16, 1, DIR>(v, t, tww);
        bar();
        if constexpr (!(ABL & 4)) F::scatter_a3(v, t, lds);
        bar();
        if constexpr (!(ABL & 4)) F::gather_a3(v, t, lds);
        if constexpr (!(ABL & 8)) dft16_tw<DIR>(&v[0], tw2f);
        bar();
        if constexpr (!(ABL & 4)) F::scatter_b3(v, t, lds);
        bar();
        if constexpr (!(ABL & 4)) F::gather_b(v, t, lds);
        if constexpr (!(ABL & 8)) dft16_tw<DIR, (DIR > 0 && R0 <= 8) ? R0 : 0>(&v[0], tw3f);
    };
    auto transform = [&](C (&v)[16]) {
        half(v, std::integral_constant<int, -1>{});
        if constexpr (!(ABL & 8)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], hreg[r]);
        }
        half(v, std::integral_constant<int, 1>{});
    };
    const unsigned G = gridDim.x;
    {
        const unsigned nwrap = nb_lo + (a.blocks - nb_hi);
        for (unsigned w = blockIdx.x; w < nwrap; w += G) {
            const unsigned b = w < nb_lo ? w : nb_hi + (w - nb_lo);
            C v[16];
            load_block(a, b, V, ut, v);
            transform(v);
            store_block<false>(a, b, V, ut, v);
        }
    }
    const unsigned total = nb_hi - nb_lo, gs = G / WPC;
    unsigned na = RA * gs, nbb = WPC >= 3 ? RB * gs : 0;
    if (na > total) na = total;
    if (na + nbb > total) nbb = total - na;
    const unsigned grp = blockIdx.x / gs;
    if (grp >= (unsigned)WPC) return;
    const unsigned lo = grp == 0 ? nb_lo : ((grp == 1 && WPC == 3) ? nb_lo + na : nb_lo + na + nbb);
    const unsigned hi = grp == 0 ? nb_lo + na : ((grp == 1 && WPC == 3) ? nb_lo + na + nbb : nb_hi);
    // this workgroup's run: a balanced partition of [lo, hi) over the gs workgroups of the group
    const unsigned w = blockIdx.x - grp * gs, n = hi - lo;
    const unsigned b0 = lo + (unsigned)((unsigned long long)w * n / gs), b1 = lo + (unsigned)((unsigned long long)(w + 1) * n / gs);
    if (b0 >= b1) return;
    C keep[R0];
    {
        const C* xb = a.x + ((long long)b0 * V + a.in_off);
#pragma unroll
        for (int r = 0; r < R0; ++r) {
            if constexpr (!(ABL & 1)) keep[r] = xb[ut + 256u * r]; else asm volatile("" : "=v"(keep[r]));
        }
    }
    for (unsigned b = b0; b < b1; ++b) {
        C v[16];
#pragma unroll
        for (int r = 0; r < R0; ++r) v[r] = keep[r];
        if constexpr (!(ABL & 1)) {
            const C* xb = a.x + ((long long)b * V + a.in_off);
#pragma unroll
            for (int r = R0; r < 16; ++r) v[r] = xb[ut + 256u * r];
        } else {
#pragma unroll
            for (int r = R0; r < 16; ++r) asm volatile("" : "=v"(v[r]));
        }
#pragma unroll
        for (int r = 0; r < R0; ++r) keep[r] = v[16 - R0 + r];
        transform(v);
        if constexpr (!(ABL & 2)) {
            C* yb = a.y + ((long long)b * V - 256 * R0);
#pragma unroll
            for (int r = R0; r < 16; ++r) yb[ut + 256u * r] = v[r];
        } else {
#pragma unroll
            for (int r = R0; r < 16; ++r) asm volatile("" : : "v"(v[r]));
        }
    }
}


// ------------------------------------------------------------------------------------------- v6 (round 3): 8192-point blocks
// 8192 = 2 x 4096 on 256 threads x 32 points: a radix-2 decimation-in-frequency layer in registers (rows r and r + 16 of
// a thread), then TWO independent 4096-point transforms (even and odd bins) whose exchanges are interleaved through two
// LDS buffers (the same four barriers per direction serve both), x H, two independent inverse transforms, a radix-2
// decimation-in-time layer in FMA form.  7168 of 8192 outputs are valid at 1024 taps (87.5 % against 75 %): per
// OUTPUT -12 % instructions, -14 % LDS traffic, -14 % loads.  The price: 64 + 64 registers of data + filter spectrum,
// i.e. 2 workgroups (2 waves per SIMD) per CU.
template <int R0, int WPC, int RA>
__global__ __launch_bounds__(256, WPC) void k_v6(Args a, unsigned nb_lo, unsigned nb_hi)
{
    constexpr int L8 = 8192;
    using F = WgFft<float, L, 256>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* ldsA = reinterpret_cast<C*>(smem_raw);
    C* ldsB = ldsA + F::LDS_ELEMS3;
    const int t = threadIdx.x;
    const unsigned ut = t;
    const unsigned V = a.V; // 8192 - 256 R0
    const float hscale = 1.0f / L8;
    auto tww = [&](int mm) { return a.wtab[mm]; };
    C ha[16], hb[16], tw2f[8], tw3f[8], w8[16];
    F::template load_twiddles16_fma<16>(tw2f, t, tww);
    F::template load_twiddles16_fma<256>(tw3f, t, tww);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned k = ut + 256u * r; // bin of the 4096-point transforms: even bins 2k, odd bins 2k + 1
        const C he = a.hs[2 * k], ho = a.hs[2 * k + 1];
        ha[r] = C{he.x * hscale, he.y * hscale};
        hb[r] = C{ho.x * hscale, ho.y * hscale};
        w8[r] = a.wtab8[k];
    }
    auto xform2 = [&](C (&p)[16], C (&q)[16], auto D) {
        constexpr int DIR = decltype(D)::value;
        F::template compute<16, 1, DIR>(p, t, tww);
        F::template compute<16, 1, DIR>(q, t, tww);
        __syncthreads();
        F::scatter_a3(p, t, ldsA);
        F::scatter_a3(q, t, ldsB);
        __syncthreads();
        F::gather_a3(p, t, ldsA);
        F::gather_a3(q, t, ldsB);
        dft16_tw<DIR>(&p[0], tw2f);
        dft16_tw<DIR>(&q[0], tw2f);
        __syncthreads();
        F::scatter_b3(p, t, ldsA);
        F::scatter_b3(q, t, ldsB);
        __syncthreads();
        F::gather_b(p, t, ldsA);
        F::gather_b(q, t, ldsB);
        dft16_tw<DIR>(&p[0], tw3f);
        dft16_tw<DIR>(&q[0], tw3f);
    };
    auto transform = [&](C (&x0)[16], C (&x1)[16]) {
        // decimation in frequency: s = x0 + x1 (even bins), d = (x0 - x1) w8192^n (odd bins)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const C sm = cadd(x0[r], x1[r]), df = csub(x0[r], x1[r]);
            x0[r] = sm;
            x1[r] = cmul(df, w8[r]);
        }
        xform2(x0, x1, std::integral_constant<int, -1>{});
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x0[r] = cmul(x0[r], ha[r]);
            x1[r] = cmul(x1[r], hb[r]);
        }
        xform2(x0, x1, std::integral_constant<int, 1>{});
        // decimation in time: z[n] = s + conj(w^n) d, z[n + 4096] = 2 s - z[n]; rows r < R0 of z[n] are discarded
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < R0) bf_tw<1, false, false>(x0[r], x1[r], w8[r]);
            else bf_tw<1, false, true>(x0[r], x1[r], w8[r]);
        }
    };
    auto load_any = [&](unsigned b, C (&x0)[16], C (&x1)[16]) {
        long long base = (long long)b * V + a.in_off;
        long long sb = base % (long long)a.n;
        if (sb < 0) sb += a.n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x0[r] = a.x[((unsigned long long)sb + ut + 256u * r) % a.n];
            x1[r] = a.x[((unsigned long long)sb + ut + 256u * (r + 16)) % a.n];
        }
    };
    auto store_any = [&](unsigned b, const C (&x0)[16], const C (&x1)[16]) {
        const long long obase = (long long)b * V - 256 * R0;
        long long room = (long long)a.n - obase;
        unsigned lim = room <= 0 ? 0u : (room > L8 ? (unsigned)L8 : (unsigned)room);
        C* yb = a.y + obase;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned n0 = ut + 256u * r, n1 = n0 + 4096u;
            if (r >= R0 && n0 < lim) yb[n0] = x0[r];
            if (n1 < lim) yb[n1] = x1[r];
        }
    };
    const unsigned G = gridDim.x;
    {
        const unsigned nwrap = nb_lo + (a.blocks - nb_hi);
        for (unsigned w = blockIdx.x; w < nwrap; w += G) {
            const unsigned b = w < nb_lo ? w : nb_hi + (w - nb_lo);
            C x0[16], x1[16];
            load_any(b, x0, x1);
            transform(x0, x1);
            store_any(b, x0, x1);
        }
    }
    const unsigned total = nb_hi - nb_lo, gs = G / WPC;
    unsigned na = RA * gs;
    if (na > total) na = total;
    const unsigned grp = blockIdx.x / gs;
    if (grp >= (unsigned)WPC) return;
    const unsigned lo = grp == 0 ? nb_lo : nb_lo + na;
    const unsigned hi = (grp == 0 && WPC > 1) ? nb_lo + na : nb_hi;
    const unsigned w2 = xcd_contiguous(blockIdx.x - grp * gs, gs);
    for (unsigned b = lo + w2; b < hi; b += gs) {
        const C* xb = a.x + ((long long)b * V + a.in_off);
        C* yb = a.y + ((long long)b * V - 256 * R0);
        C x0[16], x1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x0[r] = xb[ut + 256u * r];
            x1[r] = xb[ut + 256u * (r + 16)];
        }
        transform(x0, x1);
#pragma unroll
        for (int r = R0; r < 16; ++r) yb[ut + 256u * r] = x0[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) yb[ut + 256u * (r + 16)] = x1[r];
    }
}

// ------------------------------------------------------------------------------------------- wave-per-block
// ONE WAVE per 4096-point block, 64 points per lane: 4096 = 64 x 64, each 64-point transform entirely in the lane's
// registers (4 x 16), ONE twiddle layer (w4096^(lane k)) and ONE transposition through LDS per transform -- two
// exchanges per block instead of four, no workgroup barrier at all, four independent waves per CU (one per SIMD).
// Tables in LDS: TW[rho][lane] = w4096^(lane f(rho)), HT[rho][lane] = H'[lane + 64 f(rho)] / 4096, where register rho of
// a finished 64-point transform holds frequency f(rho) = (rho >> 4) + 4 (rho & 15).  The transposition goes through a
// 32-row buffer per wave in two halves (lanes 0..31 write, all read; lanes 32..63 write, all read): 4 x 16.5 KB + 64 KB
// of tables = 130 KB.
__device__ static constexpr float W64_TAB[64][2] = {
    {1.000000000e+00f, -0.000000000e+00f}, {9.951847267e-01f, -9.801714033e-02f}, {9.807852804e-01f, -1.950903220e-01f}, {9.569403357e-01f, -2.902846773e-01f},
    {9.238795325e-01f, -3.826834324e-01f}, {8.819212643e-01f, -4.713967368e-01f}, {8.314696123e-01f, -5.555702330e-01f}, {7.730104534e-01f, -6.343932842e-01f},
    {7.071067812e-01f, -7.071067812e-01f}, {6.343932842e-01f, -7.730104534e-01f}, {5.555702330e-01f, -8.314696123e-01f}, {4.713967368e-01f, -8.819212643e-01f},
    {3.826834324e-01f, -9.238795325e-01f}, {2.902846773e-01f, -9.569403357e-01f}, {1.950903220e-01f, -9.807852804e-01f}, {9.801714033e-02f, -9.951847267e-01f},
    {6.123233996e-17f, -1.000000000e+00f}, {-9.801714033e-02f, -9.951847267e-01f}, {-1.950903220e-01f, -9.807852804e-01f}, {-2.902846773e-01f, -9.569403357e-01f},
    {-3.826834324e-01f, -9.238795325e-01f}, {-4.713967368e-01f, -8.819212643e-01f}, {-5.555702330e-01f, -8.314696123e-01f}, {-6.343932842e-01f, -7.730104534e-01f},
    {-7.071067812e-01f, -7.071067812e-01f}, {-7.730104534e-01f, -6.343932842e-01f}, {-8.314696123e-01f, -5.555702330e-01f}, {-8.819212643e-01f, -4.713967368e-01f},
    {-9.238795325e-01f, -3.826834324e-01f}, {-9.569403357e-01f, -2.902846773e-01f}, {-9.807852804e-01f, -1.950903220e-01f}, {-9.951847267e-01f, -9.801714033e-02f},
    {-1.000000000e+00f, -1.224646799e-16f}, {-9.951847267e-01f, 9.801714033e-02f}, {-9.807852804e-01f, 1.950903220e-01f}, {-9.569403357e-01f, 2.902846773e-01f},
    {-9.238795325e-01f, 3.826834324e-01f}, {-8.819212643e-01f, 4.713967368e-01f}, {-8.314696123e-01f, 5.555702330e-01f}, {-7.730104534e-01f, 6.343932842e-01f},
    {-7.071067812e-01f, 7.071067812e-01f}, {-6.343932842e-01f, 7.730104534e-01f}, {-5.555702330e-01f, 8.314696123e-01f}, {-4.713967368e-01f, 8.819212643e-01f},
    {-3.826834324e-01f, 9.238795325e-01f}, {-2.902846773e-01f, 9.569403357e-01f}, {-1.950903220e-01f, 9.807852804e-01f}, {-9.801714033e-02f, 9.951847267e-01f},
    {-1.836970199e-16f, 1.000000000e+00f}, {9.801714033e-02f, 9.951847267e-01f}, {1.950903220e-01f, 9.807852804e-01f}, {2.902846773e-01f, 9.569403357e-01f},
    {3.826834324e-01f, 9.238795325e-01f}, {4.713967368e-01f, 8.819212643e-01f}, {5.555702330e-01f, 8.314696123e-01f}, {6.343932842e-01f, 7.730104534e-01f},
    {7.071067812e-01f, 7.071067812e-01f}, {7.730104534e-01f, 6.343932842e-01f}, {8.314696123e-01f, 5.555702330e-01f}, {8.819212643e-01f, 4.713967368e-01f},
    {9.238795325e-01f, 3.826834324e-01f}, {9.569403357e-01f, 2.902846773e-01f}, {9.807852804e-01f, 1.950903220e-01f}, {9.951847267e-01f, 9.801714033e-02f}};

constexpr int WSTR = 66;
__device__ __forceinline__ constexpr int f64map(int rho) { return (rho >> 4) + 4 * (rho & 15); }

template <int DIR>
__device__ __forceinline__ void dft64(C* v)
{
#pragma unroll
    for (int i2 = 0; i2 < 16; ++i2) dft4<DIR>(v[i2], v[16 + i2], v[32 + i2], v[48 + i2]);
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
        for (int i2 = 1; i2 < 16; ++i2) {
            const int m = i2 * k1;
            if (m == 16) v[16 * k1 + i2] = rot_i<DIR>(v[16 * k1 + i2]);
            else if (m == 8) v[16 * k1 + i2] = mul_w8_1<DIR>(v[16 * k1 + i2]);
            else if (m == 24) v[16 * k1 + i2] = mul_w8_3<DIR>(v[16 * k1 + i2]);
            else v[16 * k1 + i2] = twmul<DIR>(v[16 * k1 + i2], C{W64_TAB[m][0], W64_TAB[m][1]});
        }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft16<DIR>(&v[16 * k1]);
}

// v[rho] = A[lane, f(rho)]  ->  o[r] = A[r, lane]
__device__ __forceinline__ void wave_transpose(const C (&v)[64], C (&o)[64], int lane, C* buf)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
            C* row = buf + (lane & 31) * WSTR;
#pragma unroll
            for (int rho = 0; rho < 64; ++rho)
                if ((rho & 16) == 0) { // registers rho and rho + 16 hold frequencies f and f + 1 (f even)
                    f4 pr = {v[rho][0], v[rho][1], v[rho + 16][0], v[rho + 16][1]};
                    *reinterpret_cast<f4*>(row + f64map(rho)) = pr;
                }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 32; ++r) o[32 * h + r] = buf[r * WSTR + lane];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }
}

// natural-order input v[r] = x[lane + 64 r]; output v[rho] = X[lane + 64 f(rho)]
template <int DIR>
__device__ __forceinline__ void wave_fft4096(C (&v)[64], int lane, const C* TW, C* buf)
{
    dft64<DIR>(v);
#pragma unroll
    for (int rho = 1; rho < 64; ++rho) v[rho] = twmul<DIR>(v[rho], TW[rho * 64 + lane]);
    C o[64];
    wave_transpose(v, o, lane, buf);
    dft64<DIR>(o);
#pragma unroll
    for (int rho = 0; rho < 64; ++rho) v[rho] = o[rho];
}

template <int OVROWS, bool PREFETCH>
__global__ __launch_bounds__(256, 1) void k_wave(Args a, unsigned nb_lo, unsigned nb_hi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* TW = reinterpret_cast<C*>(smem_raw);
    C* HT = TW + 4096;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    C* buf = HT + 4096 + wave * (32 * WSTR);
    const unsigned V = a.V, ov = (unsigned)a.ov;
    unsigned long long c0 = 0, r0c = 0;
    if (a.clk && t == 0) { c0 = __builtin_readcyclecounter(); asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0c)); }
    for (int idx = t; idx < 4096; idx += 256) {
        const int rho = idx >> 6, l = idx & 63, f = f64map(rho);
        TW[idx] = a.wtab[(l * f) & (L - 1)];
        const C h = a.hs[l + 64 * f];
        HT[idx] = C{h[0] * (1.0f / L), h[1] * (1.0f / L)};
    }
    __syncthreads();
    auto transform = [&](C (&v)[64]) {
        wave_fft4096<-1>(v, lane, TW, buf);
        C u[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int rho = 16 * (i & 3) + (i >> 2); // the register that holds frequency lane + 64 i
            u[i] = cmul(v[rho], HT[rho * 64 + lane]);
        }
        wave_fft4096<1>(u, lane, TW, buf);
#pragma unroll
        for (int rho = 0; rho < 64; ++rho) v[rho] = u[rho];
    };
    const unsigned GW = gridDim.x * 4;
    // ---- edge blocks (window wraps around the vector, or outputs run past its end): general code, first
    {
        const unsigned nedge = nb_lo + (a.blocks - nb_hi);
        for (unsigned e = blockIdx.x * 4 + wave; e < nedge; e += GW) {
            const unsigned b = e < nb_lo ? e : nb_hi + (e - nb_lo);
            long long sb = ((long long)b * V + a.in_off) % (long long)a.n;
            if (sb < 0) sb += a.n;
            C v[64];
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                unsigned i = (unsigned)sb + lane + 64u * r; // < 2 n (n >= 4096 here)
                if (i >= a.n) i -= a.n;
                v[r] = a.x[i];
            }
            transform(v);
            const long long obase = (long long)b * V - ov;
            const long long room = (long long)a.n - obase;
            unsigned lim = room <= 0 ? 0u : (room > L ? (unsigned)L : (unsigned)room);
            if (lim > ov + V) lim = ov + V;
            C* yb = a.y + obase;
#pragma unroll
            for (int rho = 0; rho < 64; ++rho) {
                const unsigned np = lane + 64u * f64map(rho);
                if (np >= ov && np < lim) yb[np] = v[rho];
            }
        }
    }
    // ---- interior blocks: the four waves of a workgroup take four consecutive blocks (their windows share L2 lines)
    const unsigned slot = xcd_contiguous(blockIdx.x, gridDim.x);
    if constexpr (PREFETCH) {
        unsigned b = nb_lo + 4 * slot + wave;
        C nxt[64];
        if (b < nb_hi) {
            const C* xb = a.x + ((long long)b * V + a.in_off);
#pragma unroll
            for (int r = 0; r < 64; ++r) nxt[r] = xb[lane + 64 * r];
        }
        for (; b < nb_hi; b += GW) {
            C* yb = a.y + ((long long)b * V - ov);
            C v[64];
#pragma unroll
            for (int r = 0; r < 64; ++r) v[r] = nxt[r];
            // the next block's loads go out now (the last block re-reads its own window: no branch in the loop body)
            const unsigned bn = b + GW < nb_hi ? b + GW : b;
            const C* xn = a.x + ((long long)bn * V + a.in_off);
#pragma unroll
            for (int r = 0; r < 64; ++r) nxt[r] = xn[lane + 64 * r];
            transform(v);
#pragma unroll
            for (int rho = 0; rho < 64; ++rho)
                if (f64map(rho) >= OVROWS) yb[lane + 64 * f64map(rho)] = v[rho];
        }
    } else
    for (unsigned b = nb_lo + 4 * slot + wave; b < nb_hi; b += GW) {
        const C* xb = a.x + ((long long)b * V + a.in_off);
        C* yb = a.y + ((long long)b * V - ov);
        C v[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r] = xb[lane + 64 * r];
        transform(v);
#pragma unroll
        for (int rho = 0; rho < 64; ++rho)
            if (f64map(rho) >= OVROWS) yb[lane + 64 * f64map(rho)] = v[rho];
    }
    if (a.clk && t == 0) {
        unsigned long long c1 = __builtin_readcyclecounter(), r1;
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.clk[4 * blockIdx.x] = r0c; a.clk[4 * blockIdx.x + 1] = r1; a.clk[4 * blockIdx.x + 2] = c1 - c0; a.clk[4 * blockIdx.x + 3] = xcc & 7;
    }
}

// ------------------------------------------------------------------------------------------- host
static std::vector<std::complex<double>> fft_host(std::vector<std::complex<double>> v)
{
    const size_t n = v.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(v[i], v[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                std::complex<double> w(std::cos(ang * k), std::sin(ang * k));
                auto u = v[i + k], x = v[i + k + len / 2] * w;
                v[i + k] = u + x;
                v[i + k + len / 2] = u - x;
            }
    }
    return v;
}

int main(int argc, char** argv)
{
    const unsigned n = argc > 1 ? (unsigned)atol(argv[1]) : (1u << 24);
    const int m = argc > 2 ? atoi(argv[2]) : 1024;
    const char* only = argc > 3 ? argv[3] : "";
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    std::mt19937_64 rng(20260102);
    std::uniform_real_distribution<float> ux(-10.f, 10.f), uh(-1.f, 1.f);
    std::vector<C> hx(n), htaps(m);
    for (auto& c : hx) c = C{ux(rng), ux(rng)};
    for (auto& c : htaps) c = C{uh(rng) / m, uh(rng) / m};
    C* dx[3];
    for (auto& p : dx) { CK(hipMalloc(&p, sizeof(C) * n)); }
    CK(hipMemcpy(dx[0], hx.data(), sizeof(C) * n, hipMemcpyHostToDevice));
    {
        std::vector<C> other(n);
        for (int k = 1; k < 3; ++k) {
            for (auto& c : other) c = C{ux(rng), ux(rng)};
            CK(hipMemcpy(dx[k], other.data(), sizeof(C) * n, hipMemcpyHostToDevice));
        }
    }
    C* dy;
    CK(hipMalloc(&dy, sizeof(C) * n));
    // twiddle table
    std::vector<C> hw(L);
    for (int k = 0; k < L; ++k) {
        long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / L;
        hw[k] = C{(float)cosl(ang), (float)sinl(ang)};
    }
    C* dw;
    CK(hipMalloc(&dw, sizeof(C) * L));
    CK(hipMemcpy(dw, hw.data(), sizeof(C) * L, hipMemcpyHostToDevice));
    // filter spectra: plain (ov = m-1) and delayed (ov = round_up(m-1, 16))
    auto spectrum = [&](int delay) {
        std::vector<std::complex<double>> z(L);
        for (int k = 0; k < m; ++k) z[k + delay] = std::complex<double>(htaps[k][0], htaps[k][1]);
        auto s = fft_host(z);
        std::vector<C> f(L);
        for (int k = 0; k < L; ++k) f[k] = C{(float)s[k].real(), (float)s[k].imag()};
        C* d;
        CK(hipMalloc(&d, sizeof(C) * L));
        CK(hipMemcpy(d, f.data(), sizeof(C) * L, hipMemcpyHostToDevice));
        return d;
    };
    const int ov_plain = m - 1, ov_al = (m - 1 + 15) & ~15;
    C* hs_plain = spectrum(0);
    C* hs_al = spectrum(ov_al - ov_plain);
    C* hs_r0 = spectrum(((m - 1 + 255) & ~255) - ov_plain);
    // 8192-point blocks (k_v6): the taps delayed to a row boundary, their 8192-point spectrum, the 8192-entry twiddle table
    C *hs8 = nullptr, *dw8 = nullptr;
    {
        const int L8 = 8192, delay = ((m - 1 + 255) & ~255) - ov_plain;
        std::vector<std::complex<double>> z(L8);
        for (int k = 0; k < m; ++k) z[k + delay] = std::complex<double>(htaps[k][0], htaps[k][1]);
        auto sp = fft_host(z);
        std::vector<C> f(L8), w(L8);
        for (int k = 0; k < L8; ++k) {
            f[k] = C{(float)sp[k].real(), (float)sp[k].imag()};
            long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / L8;
            w[k] = C{(float)cosl(ang), (float)sinl(ang)};
        }
        CK(hipMalloc(&hs8, sizeof(C) * L8)); CK(hipMemcpy(hs8, f.data(), sizeof(C) * L8, hipMemcpyHostToDevice));
        CK(hipMalloc(&dw8, sizeof(C) * L8)); CK(hipMemcpy(dw8, w.data(), sizeof(C) * L8, hipMemcpyHostToDevice));
    }

    // reference outputs at selected positions: y[i] = sum_k x[(i + ceil(m/2) - 1 - k) mod n] h[k]
    std::vector<unsigned> pos;
    for (unsigned i = 0; i < 40; ++i) { pos.push_back(i); pos.push_back(n - 1 - i); }
    for (unsigned b = 1; b < 6; ++b) for (int d = -3; d <= 3; ++d) pos.push_back(b * 3072u + d);
    for (unsigned b = 1; b < 6; ++b) for (int d = -3; d <= 3; ++d) pos.push_back(b * 3073u + d);
    for (int k = 0; k < 200; ++k) pos.push_back((unsigned)(rng() % n));
    std::vector<std::complex<double>> ref(pos.size());
    const long long c = m - m / 2;
    for (size_t p = 0; p < pos.size(); ++p) {
        std::complex<double> acc = 0;
        for (int k = 0; k < m; ++k) {
            long long j = ((long long)pos[p] + c - 1 - k) % (long long)n;
            if (j < 0) j += n;
            acc += std::complex<double>(hx[j][0], hx[j][1]) * std::complex<double>(htaps[k][0], htaps[k][1]);
        }
        ref[p] = acc;
    }
    double refnorm = 0;
    for (auto& r : ref) refnorm += std::norm(r);

    struct Variant { const char* name; const void* fn; bool aligned; int per_cu; size_t lds; int r0 = 0; int l8 = 0; };
    const size_t lds_base = (size_t)(WgFft<float, L, 256>::LDS_ELEMS + 16 * 17) * sizeof(C);
    const size_t lds_l3 = (size_t)(WgFft<float, L, 256>::LDS_ELEMS3 + 16 * 17) * sizeof(C);
    const size_t lds_pf2 = (size_t)(WgFft<float, L, 256>::LDS_ELEMS3 + 2048 + 128) * sizeof(C);
    const size_t lds_dif = (size_t)(CROSS_ELEMS + 4 * PRIV_ELEMS) * sizeof(C);
    std::vector<Variant> vars = {
        {"base", (const void*)k_base<false>, false, 3, lds_base},
        {"base+as", (const void*)k_base<true>, true, 3, lds_base},
        {"ub0", (const void*)k_ub<0>, false, 3, lds_base},
        {"ub1", (const void*)k_ub<1>, false, 3, lds_base},
        {"ub2", (const void*)k_ub<2>, false, 3, lds_base},
        {"ub3", (const void*)k_ub<3>, false, 3, lds_base},
        {"ub4", (const void*)k_ub<4>, false, 3, lds_base},
        {"ub8", (const void*)k_ub<8>, false, 3, lds_base},
        {"ub16", (const void*)k_ub<16>, false, 3, lds_base},
        {"ub31", (const void*)k_ub<31>, false, 3, lds_base},
        {"dif", (const void*)k_dif<false, 2>, false, 2, lds_dif},
        {"dif+as", (const void*)k_dif<true, 2>, true, 2, lds_dif},
        {"v2 st tw0 3", (const void*)k_v2<0, 4, 0, 0, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3", (const void*)k_v2<0, 4, 1, 0, 3>, true, 3, lds_base, 4},
        {"v2 st tw3 2", (const void*)k_v2<0, 4, 3, 0, 2>, true, 2, lds_base, 4},
        {"v2 st tw3 2 db", (const void*)k_v2<0, 4, 3, 1, 2>, true, 2, lds_base, 4},
        {"v2 st tw0 2 db", (const void*)k_v2<0, 4, 0, 1, 2>, true, 2, lds_base, 4},
        {"v2 st tw0 4", (const void*)k_v2<0, 4, 0, 0, 4>, true, 4, lds_base, 4},
        {"v2 st tw0 4 dyn", (const void*)k_v2<0, 4, 0, 2, 4>, true, 4, lds_base, 4},
        {"v2 st tw0 3 dyn", (const void*)k_v2<0, 4, 0, 2, 3>, true, 3, lds_base, 4},
        {"v2 st tw0 3 dyn1", (const void*)k_v2<0, 4, 0, 3, 3>, true, 3, lds_base, 4},
        {"v2 st tw3 2 dyn", (const void*)k_v2<0, 4, 3, 2, 2>, true, 2, lds_base, 4},
        {"v2 dif 2 dyn", (const void*)k_v2<1, 4, 0, 2, 2>, true, 2, lds_dif, 4},
        {"v2 st tw3 2 skew52", (const void*)k_v2<0, 4, 3, 52, 2>, true, 2, lds_base, 4},
        {"v2 st tw3 2 skew54", (const void*)k_v2<0, 4, 3, 54, 2>, true, 2, lds_base, 4},
        {"v2 st tw3 2 skew55", (const void*)k_v2<0, 4, 3, 55, 2>, true, 2, lds_base, 4},
        {"v2 st tw3 2 skew56", (const void*)k_v2<0, 4, 3, 56, 2>, true, 2, lds_base, 4},
        {"v2 st tw3 2 skew58", (const void*)k_v2<0, 4, 3, 58, 2>, true, 2, lds_base, 4},
        {"v2 st tw1 3 skew 40/33", (const void*)k_v2<0, 4, 1, 1000 + 4033, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 skew 38/33", (const void*)k_v2<0, 4, 1, 1000 + 3833, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 skew 42/33", (const void*)k_v2<0, 4, 1, 1000 + 4233, 3>, true, 3, lds_base, 4},
        {"v2 st tw0 3 skew 40/33", (const void*)k_v2<0, 4, 0, 1000 + 4033, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 9/7", (const void*)k_v2<0, 4, 1, 2000 + 907, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 9/8", (const void*)k_v2<0, 4, 1, 2000 + 908, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 10/7", (const void*)k_v2<0, 4, 1, 2000 + 1007, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 8/7", (const void*)k_v2<0, 4, 1, 2000 + 807, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 10/6", (const void*)k_v2<0, 4, 1, 2000 + 1006, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 10/8", (const void*)k_v2<0, 4, 1, 2000 + 1008, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 9/9", (const void*)k_v2<0, 4, 1, 2000 + 909, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 10/9", (const void*)k_v2<0, 4, 1, 2000 + 1009, 3>, true, 3, lds_base, 4},
        {"v2 st tw1 3 rounds 11/8", (const void*)k_v2<0, 4, 1, 2000 + 1108, 3>, true, 3, lds_base, 4},
        {"v2 st tw0 3 rounds 9/8", (const void*)k_v2<0, 4, 0, 2000 + 908, 3>, true, 3, lds_base, 4},
        {"v2 st tw0 4 rounds 8/7/4", (const void*)k_v2<0, 4, 0, 1000000 + 80704, 4>, true, 4, lds_base, 4},
        {"v2 st tw0 4 rounds 7/6/5", (const void*)k_v2<0, 4, 0, 1000000 + 70605, 4>, true, 4, lds_base, 4},
        {"v2 st tw0 4 rounds 8/6/4", (const void*)k_v2<0, 4, 0, 1000000 + 80604, 4>, true, 4, lds_base, 4},
        {"v2 st tw0 4 rounds 9/7/4", (const void*)k_v2<0, 4, 0, 1000000 + 90704, 4>, true, 4, lds_base, 4},
        {"v2 st tw0 4 rounds 6/6/5", (const void*)k_v2<0, 4, 0, 1000000 + 60605, 4>, true, 4, lds_base, 4},
        {"v2 L3 tw0 3 rounds 9/8", (const void*)k_v2<2, 4, 0, 2000 + 908, 3>, true, 3, lds_l3, 4},
        {"v2 L3 prioX 3 rounds 9/8", (const void*)k_v2<2, 4, 4, 2000 + 908, 3>, true, 3, lds_l3, 4},
        {"v2 L3 prioC 3 rounds 9/8", (const void*)k_v2<2, 4, 8, 2000 + 908, 3>, true, 3, lds_l3, 4},
        {"v2 L3 prioX 3 even", (const void*)k_v2<2, 4, 4, 0, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw3full 3 rounds 9/8", (const void*)k_v2<2, 4, 2, 2000 + 908, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw3full 3 rounds 10/8", (const void*)k_v2<2, 4, 2, 2000 + 1008, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw3full hyb 9/7/4", (const void*)k_v2<2, 4, 2, 3000000 + 90704, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw3full hyb 9/8/4", (const void*)k_v2<2, 4, 2, 3000000 + 90804, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw3full hyb 8/7/4", (const void*)k_v2<2, 4, 2, 3000000 + 80704, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw3full hyb 8/7/5", (const void*)k_v2<2, 4, 2, 3000000 + 80705, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw3full hyb 9/7/3", (const void*)k_v2<2, 4, 2, 3000000 + 90703, 3>, true, 3, lds_l3, 4},
        {"v2 L3 tw0 3 even", (const void*)k_v2<2, 4, 0, 0, 3>, true, 3, lds_l3, 4},
        {"k3 full", (const void*)k_v3<4, 0, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 pf1 2wg 12", (const void*)k_v3<4, 0, 2, 12, 0, 1>, true, 2, lds_l3, 4},
        {"k3 pf1 2wg 11", (const void*)k_v3<4, 0, 2, 11, 0, 1>, true, 2, lds_l3, 4},
        {"k3 pf0 2wg 12", (const void*)k_v3<4, 0, 2, 12, 0, 0>, true, 2, lds_l3, 4},
        {"k3 pf1 3wg 9/8", (const void*)k_v3<4, 0, 3, 9, 8, 1>, true, 3, lds_l3, 4},
        {"k3 pf1 3wg 8/7", (const void*)k_v3<4, 0, 3, 8, 7, 1>, true, 3, lds_l3, 4},
        {"k3 pf1 3wg 7/7", (const void*)k_v3<4, 0, 3, 7, 7, 1>, true, 3, lds_l3, 4},
        {"k3 pf2 3wg 9/8", (const void*)k_v3<4, 0, 3, 9, 8, 2>, true, 3, lds_pf2, 4},
        {"k3 pf2 3wg 8/7", (const void*)k_v3<4, 0, 3, 8, 7, 2>, true, 3, lds_pf2, 4},
        {"k3 pf2 3wg 7/7", (const void*)k_v3<4, 0, 3, 7, 7, 2>, true, 3, lds_pf2, 4},
        {"k3 pf2 2wg 12", (const void*)k_v3<4, 0, 2, 12, 0, 2>, true, 2, lds_pf2, 4},
        {"k4 3wg 9/8", (const void*)k_v4<4, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k4 3wg 8/7", (const void*)k_v4<4, 3, 8, 7>, true, 3, lds_l3, 4},
        {"k4 3wg 7/7", (const void*)k_v4<4, 3, 7, 7>, true, 3, lds_l3, 4},
        {"k4 3wg 8/8", (const void*)k_v4<4, 3, 8, 8>, true, 3, lds_l3, 4},
        {"k4 2wg 12", (const void*)k_v4<4, 2, 12, 0>, true, 2, lds_l3, 4},
        {"k4 2wg 11", (const void*)k_v4<4, 2, 11, 0>, true, 2, lds_l3, 4},
        {"k4 1wg", (const void*)k_v4<4, 1, 30, 0>, true, 1, lds_l3, 4},
        {"k5 9/8", (const void*)k_v5<4, 0, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k5 8/7", (const void*)k_v5<4, 0, 3, 8, 7>, true, 3, lds_l3, 4},
        {"k5 10/8", (const void*)k_v5<4, 0, 3, 10, 8>, true, 3, lds_l3, 4},
        {"k5 9/7", (const void*)k_v5<4, 0, 3, 9, 7>, true, 3, lds_l3, 4},
        {"k5 7/7", (const void*)k_v5<4, 0, 3, 7, 7>, true, 3, lds_l3, 4},
        {"k5 10/7", (const void*)k_v5<4, 0, 3, 10, 7>, true, 3, lds_l3, 4},
        {"k5 10/9", (const void*)k_v5<4, 0, 3, 10, 9>, true, 3, lds_l3, 4},
        {"k5 11/8", (const void*)k_v5<4, 0, 3, 11, 8>, true, 3, lds_l3, 4},
        {"k5 11/7", (const void*)k_v5<4, 0, 3, 11, 7>, true, 3, lds_l3, 4},
        {"k5 11/9", (const void*)k_v5<4, 0, 3, 11, 9>, true, 3, lds_l3, 4},
        {"k5 12/8", (const void*)k_v5<4, 0, 3, 12, 8>, true, 3, lds_l3, 4},
        {"k5 memonly", (const void*)k_v5<4, 28, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k5 nomem", (const void*)k_v5<4, 3, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k6 2wg 6", (const void*)k_v6<4, 2, 6>, true, 2, 2 * lds_l3, 4, 1},
        {"k6 2wg 5", (const void*)k_v6<4, 2, 5>, true, 2, 2 * lds_l3, 4, 1},
        {"k6 2wg 7", (const void*)k_v6<4, 2, 7>, true, 2, 2 * lds_l3, 4, 1},
        {"k6 2wg even", (const void*)k_v6<4, 2, 4>, true, 2, 2 * lds_l3, 4, 1},
        {"k6 1wg", (const void*)k_v6<4, 1, 30>, true, 1, 2 * lds_l3, 4, 1},
        {"k3 1wg full", (const void*)k_v3<4, 0, 1, 30, 0>, true, 1, lds_l3, 4},
        {"k3 1wg nomem", (const void*)k_v3<4, 3, 1, 30, 0>, true, 1, lds_l3, 4},
        {"k3 1wg memonly", (const void*)k_v3<4, 28, 1, 30, 0>, true, 1, lds_l3, 4},
        {"k3 1wg noload", (const void*)k_v3<4, 1, 1, 30, 0>, true, 1, lds_l3, 4},
        {"k3 1wg nostore", (const void*)k_v3<4, 2, 1, 30, 0>, true, 1, lds_l3, 4},
        {"k3 2wg full", (const void*)k_v3<4, 0, 2, 12, 0>, true, 2, lds_l3, 4},
        {"k3 2wg nomem", (const void*)k_v3<4, 3, 2, 12, 0>, true, 2, lds_l3, 4},
        {"k3 wide st", (const void*)k_v3<4, 32, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 wide ld", (const void*)k_v3<4, 64, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 wide ldst", (const void*)k_v3<4, 96, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 wide ldst memonly", (const void*)k_v3<4, 96 + 28, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 wide ldst nolds", (const void*)k_v3<4, 96 + 4, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3f 144/104", (const void*)k_v3<4, 0, 3, 144, 104>, true, 3, lds_l3, 4},
        {"k3f 144/108", (const void*)k_v3<4, 0, 3, 144, 108>, true, 3, lds_l3, 4},
        {"k3f 144/112", (const void*)k_v3<4, 0, 3, 144, 112>, true, 3, lds_l3, 4},
        {"k3f 144/116", (const void*)k_v3<4, 0, 3, 144, 116>, true, 3, lds_l3, 4},
        {"k3f 144/120", (const void*)k_v3<4, 0, 3, 144, 120>, true, 3, lds_l3, 4},
        {"k3f 149/104", (const void*)k_v3<4, 0, 3, 149, 104>, true, 3, lds_l3, 4},
        {"k3f 149/108", (const void*)k_v3<4, 0, 3, 149, 108>, true, 3, lds_l3, 4},
        {"k3f 149/112", (const void*)k_v3<4, 0, 3, 149, 112>, true, 3, lds_l3, 4},
        {"k3f 149/116", (const void*)k_v3<4, 0, 3, 149, 116>, true, 3, lds_l3, 4},
        {"k3f 149/120", (const void*)k_v3<4, 0, 3, 149, 120>, true, 3, lds_l3, 4},
        {"k3f 152/104", (const void*)k_v3<4, 0, 3, 152, 104>, true, 3, lds_l3, 4},
        {"k3f 152/108", (const void*)k_v3<4, 0, 3, 152, 108>, true, 3, lds_l3, 4},
        {"k3f 152/112", (const void*)k_v3<4, 0, 3, 152, 112>, true, 3, lds_l3, 4},
        {"k3f 152/116", (const void*)k_v3<4, 0, 3, 152, 116>, true, 3, lds_l3, 4},
        {"k3f 152/120", (const void*)k_v3<4, 0, 3, 152, 120>, true, 3, lds_l3, 4},
        {"k3f 156/104", (const void*)k_v3<4, 0, 3, 156, 104>, true, 3, lds_l3, 4},
        {"k3f 156/108", (const void*)k_v3<4, 0, 3, 156, 108>, true, 3, lds_l3, 4},
        {"k3f 156/112", (const void*)k_v3<4, 0, 3, 156, 112>, true, 3, lds_l3, 4},
        {"k3f 156/116", (const void*)k_v3<4, 0, 3, 156, 116>, true, 3, lds_l3, 4},
        {"k3f 156/120", (const void*)k_v3<4, 0, 3, 156, 120>, true, 3, lds_l3, 4},
        {"k3 wraplast", (const void*)k_v3<4, 128, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl noload", (const void*)k_v3<4, 1, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl nostore", (const void*)k_v3<4, 2, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl nomem", (const void*)k_v3<4, 3, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl lfcount", (const void*)k_v3<4, 256, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl lfcount nomem", (const void*)k_v3<4, 256 + 3, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl nolds", (const void*)k_v3<4, 4, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl noldsbar", (const void*)k_v3<4, 20, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl novalu", (const void*)k_v3<4, 8, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl memonly", (const void*)k_v3<4, 28, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl membar", (const void*)k_v3<4, 12, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl valuonly", (const void*)k_v3<4, 23, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl valubar", (const void*)k_v3<4, 7, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl ldsonly", (const void*)k_v3<4, 11, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl valu+lds", (const void*)k_v3<4, 3, 3, 9, 8>, true, 3, lds_l3, 4},
        {"k3 abl memonly even", (const void*)k_v3<4, 28, 3, 7, 7>, true, 3, lds_l3, 4},
        {"k3 abl memonly 2wg", (const void*)k_v3<4, 28, 2, 11, 0>, true, 2, lds_l3, 4},
        {"k3 abl memonly 4wg", (const void*)k_v3<4, 28, 4, 5, 5>, true, 4, lds_l3, 4},
        {"v3 fma 3 rounds 9/8", (const void*)k_v2<3, 4, 0, 2000 + 908, 3>, true, 3, lds_l3, 4},
        {"v3 fma prune 3 rounds 9/8", (const void*)k_v2<3, 4, 1, 2000 + 908, 3>, true, 3, lds_l3, 4},
        {"v3 fma prune 3 rounds 10/8", (const void*)k_v2<3, 4, 1, 2000 + 1008, 3>, true, 3, lds_l3, 4},
        {"v3 fma prune 3 rounds 9/7", (const void*)k_v2<3, 4, 1, 2000 + 907, 3>, true, 3, lds_l3, 4},
        {"v3 fma prune 3 even", (const void*)k_v2<3, 4, 1, 0, 3>, true, 3, lds_l3, 4},
        {"v3 fma prune 4 rounds 8/7/4", (const void*)k_v2<3, 4, 1, 1000000 + 80704, 4>, true, 4, lds_l3, 4},
        {"v2 st tw0 3 rounds 9/7", (const void*)k_v2<0, 4, 0, 2000 + 907, 3>, true, 3, lds_base, 4},
        {"v2 st tw3 2 rounds 12", (const void*)k_v2<0, 4, 3, 2000 + 1200, 2>, true, 2, lds_base, 4},
        {"v2 st tw3 2 rounds 13", (const void*)k_v2<0, 4, 3, 2000 + 1300, 2>, true, 2, lds_base, 4},
        {"v2 st tw3 2 skew55 rt", (const void*)k_v2<0, 0, 3, 55, 2>, true, 2, lds_base, 4},
        {"v2 dif 2 skew55", (const void*)k_v2<1, 4, 0, 55, 2>, true, 2, lds_dif, 4},
        {"v2 dif 2", (const void*)k_v2<1, 4, 0, 0, 2>, true, 2, lds_dif, 4},
        {"v2 dif 2 db", (const void*)k_v2<1, 4, 0, 1, 2>, true, 2, lds_dif, 4},
        {"wave", (const void*)k_wave<16, false>, true, 1, (size_t)(8192 + 4 * 32 * WSTR) * sizeof(C), 4},
        {"wave pf", (const void*)k_wave<16, true>, true, 1, (size_t)(8192 + 4 * 32 * WSTR) * sizeof(C), 4},
    };
    // clock warm-up
    {
        Args a{dx[0], dy, hs_plain, dw, n, ov_plain, 3072u, -(long long)(m / 2), (n + 3071u) / 3072u, nullptr, nullptr};
        CK(hipFuncSetAttribute(vars[0].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_base));
        for (int i = 0; i < 2500; ++i) hipLaunchKernelGGL(k_base<false>, dim3(cus * 3), dim3(256), lds_base, 0, a);
        CK(hipDeviceSynchronize());
    }
    unsigned* dq;
    CK(hipMalloc(&dq, 4 * 32 * 9));
    CK(hipMemset(dq, 0, 4 * 32 * 9));
    unsigned long long* dclk;
    CK(hipMalloc(&dclk, 32 * 4096));
    std::vector<C> hy(n);
    for (auto& v : vars) {
        if (*only) { // a '|'-separated list of substrings of variant names
            bool hit = false;
            std::string o(only);
            size_t p0 = 0;
            while (p0 <= o.size()) {
                size_t p1 = o.find('|', p0);
                if (p1 == std::string::npos) p1 = o.size();
                if (p1 > p0 && std::string(v.name).find(o.substr(p0, p1 - p0)) != std::string::npos) hit = true;
                p0 = p1 + 1;
            }
            if (!hit) continue;
        }
        const int ov = v.r0 ? 256 * v.r0 : (v.aligned ? ov_al : ov_plain);
        const int LL = v.l8 ? 8192 : L;
        unsigned V = (unsigned)(LL - ov);
        if (V >= 16) V &= ~15u; // every block starts on a 128-byte line of the input
        Args a{dx[0], dy, v.l8 ? hs8 : (v.r0 ? hs_r0 : (v.aligned ? hs_al : hs_plain)), dw, n, ov, V, -(long long)(m / 2), (n + V - 1) / V, dq, nullptr};
        a.wtab8 = dw8;
        unsigned nb_lo = 0, nb_hi = a.blocks;
        while (nb_lo < a.blocks && (long long)nb_lo * V + a.in_off < 0) ++nb_lo;
        while (nb_hi > nb_lo && ((long long)(nb_hi - 1) * V + a.in_off + LL > (long long)n || (long long)(nb_hi - 1) * V + V > (long long)n)) --nb_hi;
        CK(hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, v.fn, 256, v.lds));
        const unsigned grid = std::min<unsigned>((unsigned)cus * v.per_cu, a.blocks);
        void* params[] = {&a, &nb_lo, &nb_hi};
        CK(hipMemset(dy, 0xff, sizeof(C) * n));
        CK(hipLaunchKernel(v.fn, dim3(grid), dim3(256), params, v.lds, 0));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy.data(), dy, sizeof(C) * n, hipMemcpyDeviceToHost));
        double err = 0;
        for (size_t p = 0; p < pos.size(); ++p) err += std::norm(std::complex<double>(hy[pos[p]][0], hy[pos[p]][1]) - ref[p]);
        size_t nans = 0;
        for (auto& cc : hy) if (!(cc[0] == cc[0]) || !(cc[1] == cc[1])) ++nans;
        // timing
        for (int i = 0; i < 300; ++i) { a.x = dx[i % 3]; CK(hipLaunchKernel(v.fn, dim3(grid), dim3(256), params, v.lds, 0)); }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipDeviceSynchronize());
        const int reps = 400;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) { a.x = dx[i % 3]; CK(hipLaunchKernel(v.fn, dim3(grid), dim3(256), params, v.lds, 0)); }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        if (const char* ls = getenv("LAB_SECONDS")) { // a long run of this variant (power / clock sampling from outside)
            const double secs = atof(ls);
            double total_ms = 0;
            long launches = 0;
            while (total_ms < secs * 1e3) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 2000; ++i) { a.x = dx[i % 3]; CK(hipLaunchKernel(v.fn, dim3(grid), dim3(256), params, v.lds, 0)); }
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                float m2;
                CK(hipEventElapsedTime(&m2, e0, e1));
                total_ms += m2; launches += 2000;
            }
            printf("    sustained over %.1f s: %.2f us per launch\n", total_ms / 1e3, total_ms * 1e3 / launches);
            fflush(stdout);
        }
        std::vector<unsigned long long> hclk(4 * grid, 0);
        CK(hipMemset(dclk, 0, 32 * 4096));
        a.clk = dclk;
        for (int i = 0; i < 7; ++i) { a.x = dx[i % 3]; CK(hipLaunchKernel(v.fn, dim3(grid), dim3(256), params, v.lds, 0)); }
        CK(hipDeviceSynchronize());
        a.clk = nullptr;
        CK(hipMemcpy(hclk.data(), dclk, 32 * grid, hipMemcpyDeviceToHost));
        double mhz = 0, dmin = 1e9, dmax = 0, dsum = 0, span = 0;
        if (hclk[1]) {
            unsigned long long t0 = ~0ull, t1 = 0;
            double xs[8] = {0}, xe[8] = {0}; int xn[8] = {0};
            for (unsigned g = 0; g < grid; ++g) {
                t0 = std::min(t0, hclk[4 * g]); t1 = std::max(t1, hclk[4 * g + 1]);
            }
            for (unsigned g = 0; g < grid; ++g) {
                double d = (hclk[4 * g + 1] - hclk[4 * g]) / 100.0;
                dmin = std::min(dmin, d); dmax = std::max(dmax, d); dsum += d;
                mhz += 100.0 * hclk[4 * g + 2] / (double)(hclk[4 * g + 1] - hclk[4 * g]);
                int x = (int)hclk[4 * g + 3];
                xs[x] += (hclk[4 * g] - t0) / 100.0; xe[x] += (hclk[4 * g + 1] - t0) / 100.0; xn[x]++;
            }
            mhz /= grid; span = (t1 - t0) / 100.0;
            if (getenv("LAB_DUMP")) {
                std::string fn = std::string(getenv("LAB_DUMP")) + "/wg_" + std::to_string((int)(&v - &vars[0])) + ".txt";
                if (FILE* f = fopen(fn.c_str(), "w")) {
                    fprintf(f, "# %s: wg start_us end_us xcc\n", v.name);
                    for (unsigned g = 0; g < grid; ++g)
                        fprintf(f, "%u %.2f %.2f %d\n", g, (hclk[4 * g] - t0) / 100.0, (hclk[4 * g + 1] - t0) / 100.0, (int)hclk[4 * g + 3]);
                    fclose(f);
                }
            }
            printf("    per-XCD mean start/end us:");
            for (int x = 0; x < 8; ++x) printf(" [%d: n=%d %.1f/%.1f]", x, xn[x], xn[x] ? xs[x] / xn[x] : 0, xn[x] ? xe[x] / xn[x] : 0);
            printf("\n");
        }
        printf("%-10s wg/CU %d (occ %d) V=%u blocks=%u: %6.2f us  %.3f of 8 TB/s  rel-L2 err %.2e  unwritten/NaN %zu | %.0f MHz, WG life min/mean/max %.1f/%.1f/%.1f us, first start -> last end %.1f us\n", v.name,
               v.per_cu, occ, V, a.blocks, us, 16.0 * n / (us * 1e-6) / 8e12, std::sqrt(err / refnorm), nans, mhz, dmin, dsum / grid, dmax, span);
    }
    return 0;
}
