// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 against v_fma_f64 on gfx950 -- is the f64 matrix pipe a
// faster home for a banded f64 contraction (interpolatef's inner region) than the vector ALU?
// Each wave runs REPS x 8 instructions on CHAINS independent accumulators; 4 waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma(double* out, int reps, double a0, double b0)
{
    d4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < 8 / CHAINS; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678) out[threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(double* out, int reps, double a0, double b0)
{
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = c;
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < 32 / CHAINS; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(a, acc[c], b);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    if (s == 12345.678) out[threadIdx.x] = s;
}

template <typename F> static float time_us(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 100.f;
}

int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    double* out; hipMalloc(&out, 4096);
    const int reps = 4000, grid = cus * 4; // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    const double simds = cus * 4.0;
    auto report = [&](const char* name, float us, double instr_per_wave, double flops_per_instr) {
        const double waves_per_simd = 4.0;
        const double instrs = instr_per_wave * waves_per_simd; // per SIMD
        printf("%-44s %9.1f us   %.1f ns per instruction per SIMD   %.1f TFLOP/s\n", name, us, us * 1e3 / instrs,
               instrs * simds * flops_per_instr / (us * 1e-6) / 1e12);
    };
    report("v_mfma_f64_16x16x4, 1 chain (dependent)", time_us([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, out, reps, 1.0, 2.0); }), reps * 8.0, 2048);
    report("v_mfma_f64_16x16x4, 2 chains", time_us([&] { hipLaunchKernelGGL(k_mfma<2>, dim3(grid), dim3(256), 0, 0, out, reps, 1.0, 2.0); }), reps * 8.0, 2048);
    report("v_mfma_f64_16x16x4, 4 chains", time_us([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, reps, 1.0, 2.0); }), reps * 8.0, 2048);
    report("v_fma_f64, 8 chains", time_us([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(256), 0, 0, out, reps, 1.0, 2.0); }), reps * 32.0, 128);
    report("v_fma_f64, 2 chains", time_us([&] { hipLaunchKernelGGL(k_fma<2>, dim3(grid), dim3(256), 0, 0, out, reps, 1.0, 2.0); }), reps * 32.0, 128);
    return 0;
}
