// Micro-benchmark: throughput of the workgroup FFT's LDS exchanges (scatter -> barrier -> gather ->
// barrier, 32 KB each way per 256-thread workgroup) with 1..4 workgroups per CU and nothing else.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I basic_dsp_amd/csrc -I include \
//         tools/ubench/lds_exchange_rate.hip -o tools/ubench/lds_exchange_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fft_core.h"
using namespace bdsp;

template <int WHICH>
__global__ __launch_bounds__(256, 3) void k_xchg(cpx<float>* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem_raw);
    using F = WgFft<float, 4096, 256>;
    cpx<float> v[16];
    const int t = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = cpx<float>{(float)(t + r), (float)(t - r)};
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        if (WHICH == 0) F::scatter<16, 1>(v, t, lds); else if (WHICH == 1) F::scatter<16, 16>(v, t, lds);
        else if (WHICH == 2) F::scatter_a(v, t, lds); else F::scatter_b(v, t, lds);
        __syncthreads();
        if (WHICH < 2) F::gather<16>(v, t, lds); else if (WHICH == 2) F::gather_a(v, t, lds); else F::gather_b(v, t, lds);
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
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k_xchg<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)k_xchg<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)k_xchg<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)k_xchg<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const char* names[4] = {"16,1 pad16", "16,16 pad16", "A pad32x1", "B pad32x2"};
    for (int which = 0; which < 4; ++which)
        for (int k = 1; k <= 4; ++k) {
            size_t lds = (size_t)(150 * 1024) / k - 1024;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto kern = which == 0 ? k_xchg<0> : which == 1 ? k_xchg<1> : which == 2 ? k_xchg<2> : k_xchg<3>;
            hipLaunchKernelGGL(kern, dim3(cus * k), dim3(256), lds, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(cus * k), dim3(256), lds, 0, out, iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("scatter<%s> wg/CU %d: %.3f ms -> %.1f ns per exchange per CU (%.1f ns per workgroup-exchange)\n",
                   names[which], k, ms, ms * 1e6 / iters / k, ms * 1e6 / iters);
        }
    return 0;
}
