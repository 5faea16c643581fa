// Micro-benchmark: the data movement of a TWO-pass 2^24-point f32 transform whose 4096-point columns are 8 points
// (64 bytes) wide, the tile (4096 x 8 points = 256 KB) held in the REGISTERS of one 1024-thread workgroup per CU
// (32 points per thread).  Is the access pattern itself fast enough to beat three 256-point passes (41-44 us each)?
//   pattern A (first pass):  read 64-byte runs at a 32 KB stride, write each column's 4096 results contiguously
//   pattern B (last pass):   read 64-byte runs at a 32 KB stride, write the same positions
// WORK dummy packed multiply-adds per value stand in for the butterflies (0 = pure movement).  Buffers ping-pong
// like the transform's (256 MB working set).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr unsigned NR = 4096, NC = 4096;

template <int PATTERN, int W, int WORK>
__global__ __launch_bounds__(128 * W) void k_tile(const f2* __restrict__ in, f2* __restrict__ out, unsigned tiles)
{
    constexpr int NT = 128 * W;             // threads; 32 points each
    const unsigned t = threadIdx.x, c = t % W, ti = t / W;
    for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        f2 v[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) v[r] = in[(size_t)(ti + 128 * r) * NC + W * tile + c];
#pragma unroll
        for (int k = 0; k < WORK; ++k)
#pragma unroll
            for (int r = 0; r < 32; ++r) v[r] = v[r] * f2{1.0001f, 0.9999f} + v[(r + 1) & 31];
        if (PATTERN == 0) {
            const unsigned cc = t / 128, tr = t % 128;
#pragma unroll
            for (int r = 0; r < 32; ++r) out[(size_t)(W * tile + cc) * NR + tr + 128 * r] = v[r];
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) out[(size_t)(ti + 128 * r) * NC + W * tile + c] = v[r];
        }
    }
    (void)NT;
}

template <typename F> static float time_us(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f(i);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) f(i);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t n = (size_t)NR * NC;
    f2* buf[2];
    for (auto& p : buf) { hipMalloc(&p, n * sizeof(f2)); hipMemset(p, 0, n * sizeof(f2)); }
#define RUN(P, W, WORK, GRIDMUL)                                                                                     \
    {                                                                                                                 \
        const unsigned tiles = NC / W, grid = (unsigned)cus * GRIDMUL < tiles ? (unsigned)cus * GRIDMUL : tiles;      \
        float us = time_us([&](int i) { hipLaunchKernelGGL((k_tile<P, W, WORK>), dim3(grid), dim3(128 * W), 0, 0, buf[i & 1], buf[(i + 1) & 1], tiles); }); \
        printf("pattern %c  %2d-point (%3d-byte) runs, %4d threads, %d workgroups per CU, work %2d: %7.1f us per pass  (%.2f TB/s moved)\n", \
               P ? 'B' : 'A', W, W * 8, 128 * W, GRIDMUL, WORK, us, 2.0 * n * 8 / (us * 1e-6) / 1e12);              \
    }
    RUN(0, 8, 0, 1) RUN(1, 8, 0, 1) RUN(0, 8, 0, 2) RUN(1, 8, 0, 2)
    RUN(0, 8, 8, 1) RUN(1, 8, 8, 1) RUN(0, 8, 16, 1) RUN(1, 8, 16, 1)
    RUN(0, 4, 0, 2) RUN(1, 4, 0, 2) RUN(0, 4, 8, 2) RUN(1, 4, 8, 2)
    RUN(0, 2, 0, 4) RUN(1, 2, 0, 4)
    return 0;
}
