// Micro-benchmark: what does it cost to move data between the REGISTER index and the LANE index without LDS?
// A radix-16 stage boundary exchanges four index bits; v_permlane32_swap / v_permlane16_swap (new on gfx950) each
// exchange ONE register-index bit with lane bit 5 / 4 for a register pair in one VALU instruction, so two of the four
// bits of an exchange could ride on them (the other two are lane bits 0..3 or wave bits and need DPP selects or LDS).
// Timed here, per 256-thread workgroup holding 16 complex f32 registers per thread (the overlap-save block's shape):
//   lds      one full exchange through LDS in the shipped layout (scatter_b + gather_b, two barriers)
//   swap2    two register bits <-> lane bits 4,5 by permlane swaps (2 x 16 instructions), no LDS, no barrier
//   dpp2     two register bits <-> lane bits 2,3 by v_cndmask + DPP row shifts (2 x 32 instructions)
// and a radix-4 butterfly group between repetitions so that the moves cannot be folded away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fft_core.h"
using namespace bdsp;
using C = cpx<float>;

__device__ __forceinline__ void swap_bit_lane5(C (&v)[16], int bit)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r & bit) continue;
        const int q = r | bit;
        // lanes 32..63 of v[r] <-> lanes 0..31 of v[q]: afterwards register bit `bit` and lane bit 5 have traded places
        auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[r].x), __builtin_bit_cast(unsigned, v[q].x), false, false);
        auto b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[r].y), __builtin_bit_cast(unsigned, v[q].y), false, false);
        v[r] = C{__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, b[0])};
        v[q] = C{__builtin_bit_cast(float, a[1]), __builtin_bit_cast(float, b[1])};
    }
}
__device__ __forceinline__ void swap_bit_lane4(C (&v)[16], int bit)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r & bit) continue;
        const int q = r | bit;
        auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v[r].x), __builtin_bit_cast(unsigned, v[q].x), false, false);
        auto b = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v[r].y), __builtin_bit_cast(unsigned, v[q].y), false, false);
        v[r] = C{__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, b[0])};
        v[q] = C{__builtin_bit_cast(float, a[1]), __builtin_bit_cast(float, b[1])};
    }
}
// register bit <-> lane bit 3 (xor 8 inside a row of 16 = row_ror:8) and lane bit 2 (row_shr:4 / row_shl:4)
template <int LANEBIT>
__device__ __forceinline__ void swap_bit_dpp(C (&v)[16], int bit, int lane)
{
    const bool hi = (lane >> LANEBIT) & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r & bit) continue;
        const int q = r | bit;
        float ax = v[r].x, ay = v[r].y, bx = v[q].x, by = v[q].y;
        // new a = hi ? b(from lane ^ m) : a ; new b = hi ? b : a(from lane ^ m)
        constexpr int m = 1 << LANEBIT;
        float pbx = __shfl_xor(bx, m, 64), pby = __shfl_xor(by, m, 64), pax = __shfl_xor(ax, m, 64), pay = __shfl_xor(ay, m, 64);
        v[r] = C{hi ? pbx : ax, hi ? pby : ay};
        v[q] = C{hi ? bx : pax, hi ? by : pay};
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_x(C* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C* lds = reinterpret_cast<C*>(smem_raw);
    using F = WgFft<float, 4096, 256>;
    const int t = threadIdx.x, lane = t & 63;
    C v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = C{(float)(t + r) * 1e-3f, (float)(t - r) * 1e-3f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            __syncthreads();
            F::scatter_b(v, t, lds);
            __syncthreads();
            F::gather_b(v, t, lds);
        } else if (MODE == 1) {
            swap_bit_lane5(v, 8);
            swap_bit_lane4(v, 4);
        } else if (MODE == 2) {
            swap_bit_dpp<3>(v, 8, lane);
            swap_bit_dpp<2>(v, 4, lane);
        }
        dft4<-1>(v[0], v[4], v[8], v[12]);
        dft4<-1>(v[1], v[5], v[9], v[13]);
        dft4<-1>(v[2], v[6], v[10], v[14]);
        dft4<-1>(v[3], v[7], v[11], v[15]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)blockIdx.x * 4096 + t + 256 * r] = v[r];
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    C* out;
    hipMalloc(&out, sizeof(C) * 4096 * cus * 3);
    const int iters = 2000;
    const char* names[4] = {"lds exchange + radix-4 group", "permlane swaps (2 bits) + radix-4 group", "shuffle selects (2 bits) + radix-4 group", "radix-4 group only"};
    const size_t lds = 40 * 1024;
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int it) {
            if (mode == 0) hipLaunchKernelGGL(k_x<0>, dim3(cus * 3), dim3(256), lds, 0, out, it);
            else if (mode == 1) hipLaunchKernelGGL(k_x<1>, dim3(cus * 3), dim3(256), lds, 0, out, it);
            else if (mode == 2) hipLaunchKernelGGL(k_x<2>, dim3(cus * 3), dim3(256), lds, 0, out, it);
            else hipLaunchKernelGGL(k_x<3>, dim3(cus * 3), dim3(256), lds, 0, out, it);
        };
        launch(200);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        launch(iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f ns per repetition per CU (3 workgroups per CU)\n", names[mode], ms * 1e6 / iters / 3);
    }
    return 0;
}
