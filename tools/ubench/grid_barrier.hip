// Micro-benchmark behind the "fuse the two passes of a 2^20-point transform into one launch" question (config C2:
// two dependent launches of ~7 us each): what does a grid-wide barrier cost in that geometry (256 workgroups of 256
// threads, one per CU, all resident) against the kernel boundary it would replace?
//   barrier   a monotonic device-scope counter: release fence, one atomic add per workgroup, relaxed polling with
//             s_sleep, acquire fence (the form MI355X_MICROARCH.md prices at ~7 us; the XCD-hierarchical one at ~4 us)
//   boundary  the same small amount of work as two dependent launches on one stream
// Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_barriers(unsigned* ctr, float* out, int rounds, unsigned base)
{
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        acc = acc * 1.0001f + 1.0f;
        out[blockIdx.x * 256 + threadIdx.x] = acc; // something to publish
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = base + (unsigned)(r + 1) * gridDim.x;
            unsigned spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        acc += out[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x] * 1e-9f; // read a neighbour's value
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_step(float* out)
{
    float acc = out[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x] * 1.0001f + 1.0f;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* ctr; float* out;
    hipMalloc(&ctr, 64); hipMemset(ctr, 0, 64);
    hipMalloc(&out, sizeof(float) * 256 * cus * 2);
    hipMemset(out, 0, sizeof(float) * 256 * cus * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_step, dim3(cus), dim3(256), 0, 0, out); // clock
    hipDeviceSynchronize();
    unsigned base = 0;
    for (int rounds : {1, 11, 101}) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_barriers, dim3(cus), dim3(256), 0, 0, ctr, out, rounds, base);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            base += (unsigned)rounds * cus;
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("one launch with %3d grid barriers (%d workgroups): %.2f us\n", rounds, cus, best * 1e3);
    }
    for (int n : {1, 2, 11, 101}) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_step, dim3(cus), dim3(256), 0, 0, out);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%3d dependent launches of a trivial kernel: %.2f us\n", n, best * 1e3);
    }
    return 0;
}
