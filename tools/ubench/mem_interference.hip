// Micro-benchmark: does global-memory traffic slow the overlap-save block's butterfly + exchange core even when
// nothing waits for it?  The core (six butterfly groups, four LDS exchanges in the shipped layouts, spectrum
// product) runs on register data; next to it the kernel issues the block's 32 KB of loads and 24 KB of stores in
// one of several forms:
//   LOADS 0 none
//         1 fire-and-forget: untracked asm loads into registers nobody reads (pure interference, no dependency)
//         2 fire-and-forget through LDS-DMA (global_load_lds_dwordx4 into a staging area nobody reads)
//         3 register prefetch at distance 1 (compiler-tracked loads into nx[], consumed by the next block)
//         4 LDS-DMA prefetch at distance 1 (the next block lands in a 32 KB staging area while this one computes;
//           the block starts with 16 ds_read_b64 from it)
//         5 no prefetch: the block's loads at the loop top (the shipped kernel's structure)
//   STORES 0 none, 1 the block's 3072 valid points
// Same persistent XCD-contiguous walk over 5462 blocks as the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "fft_core.h"
using namespace bdsp;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int COMP, int LOADS, int STORES, int WPC>
__global__ __launch_bounds__(256, WPC) void k_mix(const cpx<float>* __restrict__ x, cpx<float>* __restrict__ y,
                                                 unsigned n, unsigned blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using F = WgFft<float, 4096, 256>;
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem_raw);
    cpx<float>* stage = lds + 4096 + 272 + 64; // 32 KB staging area for the LDS-DMA modes
    const int t = threadIdx.x;
    const unsigned ut = t;
    constexpr unsigned V = 3072, ov = 1023;
    cpx<float> tw[15], h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = cpx<float>{0.5f, 0.25f * r};
#pragma unroll
    for (int r = 0; r < 15; ++r) { float s, c; sincosf(0.001f * (t + 1) * (r + 1), &s, &c); tw[r] = cpx<float>{c, s}; }

    unsigned wl = blockIdx.x;
    if ((gridDim.x & 7) == 0) wl = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const unsigned G = gridDim.x;

    auto in_base = [&](unsigned b) -> long long {
        long long base = (long long)b * V - 512;
        if (base < 0) base = 0;
        if (base + 4096 > (long long)n) base = (long long)n - 4096;
        return base;
    };
    auto load_regs = [&](unsigned b, cpx<float> (&d)[16]) {
        const cpx<float>* xb = x + in_base(b);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = xb[ut + 256u * r];
    };
    auto dma_block = [&](unsigned b) {
        const cpx<float>* xb = x + in_base(b);
        const int w = t >> 6, l = t & 63;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const cpx<float>* g = xb + ((w * 8 + q) * 128 + 2 * l);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(stage + (w * 8 + q) * 128), 16, 0, 0);
        }
    };
    auto core = [&](cpx<float> (&v)[16]) {
        if (!COMP) return;
        F::template compute_pre<16, 256, -1>(v, tw);
        __syncthreads();
        F::scatter_a(v, t, lds);
        __syncthreads();
        F::gather_a(v, t, lds);
        F::template compute_pre<16, 256, -1>(v, tw);
        __syncthreads();
        F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        F::template compute_pre<16, 256, -1>(v, tw);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], h[r]);
        F::template compute_pre<16, 256, 1>(v, tw);
        __syncthreads();
        F::scatter_a(v, t, lds);
        __syncthreads();
        F::gather_a(v, t, lds);
        F::template compute_pre<16, 256, 1>(v, tw);
        __syncthreads();
        F::scatter_b(v, t, lds);
        __syncthreads();
        F::gather_b(v, t, lds);
        F::template compute_pre<16, 256, 1>(v, tw);
    };
    auto store = [&](unsigned b, const cpx<float> (&v)[16]) {
        if (!STORES) return;
        long long ob = (long long)b * V - ov;
        if (ob + 4096 > (long long)n) ob = (long long)n - 4096;
        if (ob < 0) ob = 0;
        cpx<float>* yb = y + ob;
#pragma unroll
        for (int r = 4; r < 16; ++r) yb[ut + 256u * r] = v[r];
    };

    cpx<float> v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = cpx<float>{(float)(t + r) * 1e-3f, (float)(t - r) * 1e-3f};

    if constexpr (LOADS == 3) {
        cpx<float> nx[16];
        if (wl < blocks) load_regs(wl, nx);
        for (unsigned b = wl; b < blocks; b += G) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = nx[r];
            if (b + G < blocks) load_regs(b + G, nx);
            core(v);
            store(b, v);
        }
    } else if constexpr (LOADS == 4) {
        if (wl < blocks) dma_block(wl);
        for (unsigned b = wl; b < blocks; b += G) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = stage[ut + 256u * r];
            __syncthreads(); // everyone has read the staging area
            if (b + G < blocks) dma_block(b + G);
            core(v);
            store(b, v);
        }
    } else {
        for (unsigned b = wl; b < blocks; b += G) {
            if constexpr (LOADS == 5) load_regs(b, v);
            float2 d[16];
            if constexpr (LOADS == 1) {
                const cpx<float>* xb = x + in_base(b) + ut;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d[r]) : "v"(xb + 256 * r));
            }
            if constexpr (LOADS == 2) dma_block(b);
            core(v);
            if constexpr (LOADS == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(d[r]));
            }
            if constexpr (LOADS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store(b, v);
        }
    }
    if (!STORES || blocks == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) y[(size_t)blockIdx.x * 4096 + ut + 256u * r] = v[r];
    }
}

static int g_cus;
static cpx<float>*g_x[3], *g_y;
static const unsigned N = 1u << 24, VV = 3072, BLOCKS = (N + VV - 1) / VV;

template <int COMP, int LOADS, int STORES, int WPC>
static void run(const char* name)
{
    auto kern = k_mix<COMP, LOADS, STORES, WPC>;
    const bool dma = LOADS == 2 || LOADS == 4;
    size_t lds = (size_t)(4096 + 272 + 64) * 8 + (dma ? 32768 : 0);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds);
    int per = occ < WPC ? occ : WPC;
    if (per < 1) { printf("%-44s wg/CU %d: does not fit\n", name, WPC); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(g_cus * per), dim3(256), lds, 0, g_x[i % 3], g_y, N, BLOCKS);
    hipDeviceSynchronize();
    const int reps = 40;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(g_cus * per), dim3(256), lds, 0, g_x[i % 3], g_y, N, BLOCKS);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / reps;
    printf("%-44s wg/CU %d (occ %d): %6.1f us per launch, %5.0f ns per block per CU\n", name, per, occ, us,
           us * 1e3 / ((double)BLOCKS / g_cus));
}

int main()
{
    hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (auto& p : g_x) { hipMalloc(&p, sizeof(cpx<float>) * N); hipMemset(p, 0, sizeof(cpx<float>) * N); }
    hipMalloc(&g_y, sizeof(cpx<float>) * N);
    // warm the clock
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k_mix<1, 0, 0, 3>), dim3(g_cus * 3), dim3(256), (4096 + 272 + 64) * 8, 0, g_x[0], g_y, N, BLOCKS);
    hipDeviceSynchronize();
#define RUN(C, L, S, W, NAME) run<C, L, S, W>(NAME)
    RUN(1, 0, 0, 3, "core only");
    RUN(1, 0, 0, 2, "core only");
    RUN(0, 5, 1, 3, "memory only (loads at top + stores)");
    RUN(0, 5, 1, 2, "memory only (loads at top + stores)");
    RUN(1, 5, 1, 3, "core + loads at top + stores (shipped shape)");
    RUN(1, 5, 1, 2, "core + loads at top + stores (shipped shape)");
    RUN(1, 5, 0, 3, "core + loads at top");
    RUN(1, 0, 1, 3, "core + stores");
    RUN(1, 1, 0, 3, "core + fire-and-forget register loads");
    RUN(1, 1, 1, 3, "core + fire-and-forget register loads + stores");
    RUN(1, 2, 0, 2, "core + fire-and-forget LDS-DMA loads");
    RUN(1, 2, 1, 2, "core + fire-and-forget LDS-DMA loads + stores");
    RUN(1, 3, 0, 2, "core + register prefetch");
    RUN(1, 3, 1, 2, "core + register prefetch + stores");
    RUN(1, 3, 1, 3, "core + register prefetch + stores");
    RUN(1, 4, 0, 2, "core + LDS-DMA prefetch");
    RUN(1, 4, 1, 2, "core + LDS-DMA prefetch + stores");
    RUN(0, 4, 1, 2, "memory only, LDS-DMA prefetch + stores");
    return 0;
}
