// Micro-benchmark: the overlap-save block WITHOUT global memory traffic -- six butterfly groups and four
// LDS exchanges per "block", 1..3 workgroups per CU.  Separates the compute+LDS core from HBM effects.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fft_core.h"
using namespace bdsp;

template <int MODE> // 0 full, 1 compute only, 2 exchanges only
__global__ __launch_bounds__(256, 3) void k_core(cpx<float>* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem_raw);
    using F = WgFft<float, 4096, 256>;
    cpx<float> v[16], tw[15], h[16];
    const int t = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[r] = cpx<float>{(float)(t + r) * 1e-3f, (float)(t - r) * 1e-3f}; h[r] = cpx<float>{0.5f, 0.25f * r}; }
#pragma unroll
    for (int r = 0; r < 15; ++r) { float s, c; sincosf(0.001f * (t + 1) * (r + 1), &s, &c); tw[r] = cpx<float>{c, s}; }
    auto xchg = [&](int which) {
        if (MODE == 1) return;
        __syncthreads();
        if (which == 0) F::scatter<16, 1>(v, t, lds); else F::scatter<16, 16>(v, t, lds);
        __syncthreads();
        F::gather<16>(v, t, lds);
    };
    auto comp = [&]() { if (MODE != 2) F::compute_pre<16, 256, -1>(v, tw); };
    for (int i = 0; i < iters; ++i) {
        comp(); xchg(0); comp(); xchg(1); comp();
        if (MODE != 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], h[r]);
        }
        comp(); xchg(0); comp(); xchg(1); comp();
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)blockIdx.x * 4096 + t + 256 * r] = v[r];
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    cpx<float>* out;
    hipMalloc(&out, sizeof(cpx<float>) * 4096 * cus * 4);
    const int iters = 500;
    const void* ks[3] = {(const void*)k_core<0>, (const void*)k_core<1>, (const void*)k_core<2>};
    const char* names[3] = {"compute+exchange", "compute only", "exchange only"};
    for (int mode = 0; mode < 3; ++mode) {
        hipFuncSetAttribute(ks[mode], hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        for (int k = 1; k <= 3; ++k) {
            size_t lds = (size_t)(150 * 1024) / k - 1024;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int it) {
                if (mode == 0) hipLaunchKernelGGL(k_core<0>, dim3(cus * k), dim3(256), lds, 0, out, it);
                else if (mode == 1) hipLaunchKernelGGL(k_core<1>, dim3(cus * k), dim3(256), lds, 0, out, it);
                else hipLaunchKernelGGL(k_core<2>, dim3(cus * k), dim3(256), lds, 0, out, it);
            };
            launch(10);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            launch(iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-18s wg/CU %d: %.3f ms -> %.0f ns per block per CU\n", names[mode], k, ms, ms * 1e6 / iters / k);
        }
    }
    return 0;
}
