// Micro-benchmark: issue rate of scalar vs packed f32 VALU ops on gfx950 (one wave64 per SIMD and
// several), to decide whether FFT butterflies should be written for v_pk_*_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    f2 pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(b));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a1) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a2) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a3) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a4) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a5) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a6) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a7) : "v"(b));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p4) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p5) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p6) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p7) : "v"(pb));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p4) : "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p5) : "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p6) : "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p7) : "v"(pb));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
double run(int blocks_per_cu, int iters, float* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)grid * 4 /*waves*/ * iters * 64.0; // wave-instructions
    return insts / (ms * 1e-3);  // wave-instr per second, whole chip
}

int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const char* names[4] = {"v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32"};
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        double r[4] = {run<0>(bpc, 2000, d), run<1>(bpc, 2000, d), run<2>(bpc, 2000, d), run<3>(bpc, 2000, d)};
        for (int m = 0; m < 4; ++m) {
            double per_simd = r[m] / (256.0 * 4.0);
            printf("waves/SIMD=%d %-14s %.3e wave-instr/s/SIMD  -> %.2f clk/instr @2.4GHz, lanes-ops %.1f T/s\n", bpc, names[m], per_simd,
                   2.4e9 / per_simd, r[m] * 64 * (m >= 2 ? 2 : 1) / 1e12);
        }
    }
    return 0;
}
