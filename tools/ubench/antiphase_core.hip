// Micro-benchmark: the anti-phase two-subgroup block program (k_overlap_save_sys) WITHOUT global memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include "fft_core.h"
using namespace bdsp;

template <int SKEW>
__global__ __launch_bounds__(512, 1) void k_anti(cpx<float>* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using F = WgFft<float, 4096, 256>;
    const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int t = threadIdx.x & 255;
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem_raw) + g * F::LDS_ELEMS;
    cpx<float> v[16], tw[15], h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[r] = cpx<float>{(float)(t + r) * 1e-3f, (float)(t - r) * 1e-3f}; h[r] = cpx<float>{0.5f, 0.25f * r}; }
#pragma unroll
    for (int r = 0; r < 15; ++r) { float s, c; sincosf(0.001f * (t + 1) * (r + 1), &s, &c); tw[r] = cpx<float>{c, s}; }
    auto twm = [&](auto dir) {
        constexpr int DIR = decltype(dir)::value;
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = twmul<DIR>(v[r], tw[r - 1]);
    };
    using Fwd = std::integral_constant<int, -1>;
    using Inv = std::integral_constant<int, 1>;
    if (SKEW && g) __syncthreads();
    for (int i = 0; i < iters; ++i) {
        dft16_a<-1>(v);
        __syncthreads();
        dft16_b<-1>(v); F::scatter<16, 1>(v, t, lds);
        __syncthreads();
        F::gather<16>(v, t, lds); twm(Fwd{}); dft16_a<-1>(v);
        __syncthreads();
        dft16_b<-1>(v); F::scatter<16, 16>(v, t, lds);
        __syncthreads();
        F::gather<16>(v, t, lds); twm(Fwd{}); dft16_a<-1>(v);
        __syncthreads();
        dft16_b<-1>(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = cmul(v[r], h[r]);
        dft16_a<1>(v);
        __syncthreads();
        dft16_b<1>(v); F::scatter<16, 1>(v, t, lds);
        __syncthreads();
        F::gather<16>(v, t, lds); twm(Inv{}); dft16_a<1>(v);
        __syncthreads();
        dft16_b<1>(v); F::scatter<16, 16>(v, t, lds);
        __syncthreads();
        F::gather<16>(v, t, lds); twm(Inv{}); dft16_a<1>(v);
        __syncthreads();
        dft16_b<1>(v);
        __syncthreads();
    }
    if (SKEW && !g) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)blockIdx.x * 8192 + threadIdx.x + 512 * r] = v[r];
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    cpx<float>* out;
    hipMalloc(&out, sizeof(cpx<float>) * 8192 * cus);
    const int iters = 500;
    size_t lds = 2 * (4096 + 256) * sizeof(cpx<float>);
    hipFuncSetAttribute((const void*)k_anti<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_anti<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int skew = 0; skew < 2; ++skew) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int it) {
            if (skew) hipLaunchKernelGGL(k_anti<1>, dim3(cus), dim3(512), lds, 0, out, it);
            else hipLaunchKernelGGL(k_anti<0>, dim3(cus), dim3(512), lds, 0, out, it);
        };
        launch(10);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        launch(iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("two subgroups, skew %d: %.3f ms -> %.0f ns per block per CU (two blocks per %.0f ns)\n", skew, ms,
               ms * 1e6 / iters / 2, ms * 1e6 / iters);
    }
    return 0;
}
