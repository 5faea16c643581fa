// Micro-benchmark: do a workgroup's repeated global stores to the SAME small region stay in its XCD's L2 (write-back),
// or does every store's data leave the L2 towards the memory side (write-through)?  Decides whether an exchange
// buffer that lives in L2 can save fabric WRITE crossings or only read crossings (two-trip FFT plan).
// Each workgroup rewrites its own `bytes_per_wg` region `rounds` times (plain 16-byte stores, s_waitcnt vmcnt(0)
// between rounds) and reads it back once at the end.  Run under rocprofv3 --pmc WRITE_SIZE (and TCC_EA0_WRREQ_sum):
//   write-back    -> WRITE_SIZE ~ grid * bytes_per_wg            (one eviction / end-of-kernel flush)
//   write-through -> WRITE_SIZE ~ grid * bytes_per_wg * rounds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE> // 0 plain stores, 1 nt stores
__global__ __launch_bounds__(256) void k_rewrite(f4* buf, size_t f4_per_wg, int rounds, float* sink)
{
    f4* p = buf + (size_t)blockIdx.x * f4_per_wg;
    for (int r = 0; r < rounds; ++r) {
        const f4 v = f4{(float)r, (float)threadIdx.x, 1.0f, 2.0f};
        for (size_t i = threadIdx.x; i < f4_per_wg; i += 256) {
            if (MODE == 0) p[i] = v;
            else __builtin_nontemporal_store(v, &p[i]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float acc = 0;
    for (size_t i = threadIdx.x; i < f4_per_wg; i += 256) acc += p[i].x;
    if (acc == -1.0f) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 50;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    f4* buf; float* sink;
    hipMalloc(&buf, (size_t)cus * (1 << 20));
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
    for (size_t kb : {4, 16, 64, 256}) { // per workgroup: x 32 workgroups per XCD = 128 KB, 512 KB, 2 MB, 8 MB per XCD
        const size_t n4 = kb * 1024 / 16;
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k_rewrite<0>, dim3(cus), dim3(256), 0, 0, buf, n4, rounds, sink);
        else hipLaunchKernelGGL(k_rewrite<1>, dim3(cus), dim3(256), 0, 0, buf, n4, rounds, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s stores, %4zu KB per workgroup (%5.2f MB per XCD), %d rounds: %.1f us; stored %.1f MB in all, footprint %.1f MB\n", mode ? "nt   " : "plain", kb,
               kb * 32 / 1024.0, rounds, ms * 1e3, cus * kb * rounds / 1024.0, cus * kb / 1024.0);
    }
    return 0;
}
