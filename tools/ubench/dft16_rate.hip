// Micro-benchmark: throughput of the real butterfly code (packed-f32 dft16 + 15 twiddle multiplies from
// fft_core.h) with 1, 2, 3, 4 waves per SIMD, no LDS or memory traffic.  Answers: is the overlap-save
// kernel's butterfly phase VALU-throughput bound, and at how many cycles per v_pk_* instruction?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I basic_dsp_amd/csrc -I include \
//         tools/ubench/dft16_rate.hip -o tools/ubench/dft16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fft_core.h"
using namespace bdsp;

__global__ __launch_bounds__(256, 3) void k_dft(cpx<float>* out, int iters, long long* clk)
{
    extern __shared__ char smem[];
    using F = WgFft<float, 4096, 256>;
    cpx<float> v[16], tw[15];
    const int t = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = cpx<float>{(float)(t + r) * 1e-3f, (float)(t - r) * 1e-3f};
#pragma unroll
    for (int r = 0; r < 15; ++r) { float s, c; sincosf(0.001f * (t + 1) * (r + 1), &s, &c); tw[r] = cpx<float>{c, s}; }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        F::compute_pre<16, 256, -1>(v, tw);
        asm volatile("" ::: "memory");
    }
    long long t1 = clock64();
    if (smem[0] == 77) v[0].x += 1.0f; // keep the LDS allocation alive
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)blockIdx.x * 4096 + t + 256 * r] = v[r];
    if (t == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    cpx<float>* out; long long* clk;
    hipMalloc(&out, sizeof(cpx<float>) * 4096 * cus * 4);
    hipMalloc(&clk, 8);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k_dft, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int k = 1; k <= 3; ++k) {
        size_t lds = (size_t)(150 * 1024) / k - 1024; // k workgroups fit per CU, k+1 do not
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_dft, dim3(cus * k), dim3(256), lds, 0, out, 10, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_dft, dim3(cus * k), dim3(256), lds, 0, out, iters, clk);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        // per SIMD: k waves each run `iters` dft16+twiddle groups
        double ns_per_group = ms * 1e6 / iters / k;
        printf("waves/SIMD %d: %.3f ms, %.1f ns per (15 twmul + dft16) per SIMD, wave clock64 delta %lld (%.1f per group)\n",
               k, ms, ns_per_group, c, (double)c / iters);
    }
    return 0;
}
