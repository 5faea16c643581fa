// mixed_radix_reg2.h -- the register-resident TWO-stage mixed-radix kernel for short smooth lengths n = R0 R1 < 300 (100 = 10 10,
// 120 = 12 10, 240 = 16 15 ...); included by mixed_radix_reg3_f32.hip / _f64.hip next to mixed_radix_reg3.h, whose I/O descriptor
// (MrReg3Io) and radix set it shares.  Same idea as k_mr_reg3 with one stage less: a thread keeps one radix-R butterfly per stage
// in registers, ONE padded LDS exchange between the two stages, persistent workgroups, the stage-1 twiddles w_n^(r j) per-thread
// constants loaded once.  A transform is only R0 (the larger radix) threads wide, so a 256-thread workgroup takes 256 / R0 of them
// at once -- consecutive vectors, i.e. ONE contiguous chunk of the batch: the workgroup brings that chunk into LDS in natural
// order with unit-stride loads (k_mr_reg3's stage-0 runs would be 40-200 bytes long here), both stages work out of LDS, and the
// spectra leave the same way.  Every fused option (rotation, scale, window, real input, magnitude / real-part output) lives in those
// two rolled loops.  Replaces k_mr_wg for these lengths (table copy per workgroup, four LDS round trips, 1024 threads).
// f32 only (mixed_radix_reg3_f64.hip says why).  *Measured*: profiles/r06_plan_probe_valid.txt run 5.
#pragma once
#include "mixed_radix_reg3.h"

namespace bdsp {

template <int R0, int R1, int ESZ>
struct MrReg2 {
    static_assert(R0 >= R1, "largest radix first");
    static constexpr int N = R0 * R1;
    static constexpr int NT = R0;        // threads per transform: R1 of them in stage 0, all R0 in stage 1
    static constexpr int SA = R0 | 1;    // exchange row stride (odd: the 64 lanes of a store hit distinct banks)
    static constexpr int LA = R1 * SA;
    // 256 threads unless their transforms' two buffers would exceed 48 KB of LDS (f64): then 128, so that three or four
    // workgroups still share a CU (*measured* f64 65536 x 100 points with 84 KB per workgroup: one workgroup per CU, 100 us
    // against 80 for k_mr_wg)
    static constexpr int THREADS = (256 / NT) * (N + LA) * 2 * ESZ > 48 * 1024 ? 128 : 256;
    static constexpr int B = THREADS / NT; // transforms per workgroup
};

template <typename T, int DIR, int R0, int R1>
__global__ __launch_bounds__((MrReg2<R0, R1, (int)sizeof(T)>::THREADS)) void k_mr_reg2(MrReg3Io<T> io, const cpx<T>* __restrict__ wtab,
                                                                                        unsigned long long batch)
{
    using P = MrReg2<R0, R1, (int)sizeof(T)>;
    constexpr int N = P::N, CH = N * P::B, TH = P::THREADS; // CH: one workgroup's chunk, B consecutive vectors
    constexpr int PER = (CH + TH - 1) / TH;                 // its elements per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* nat = reinterpret_cast<cpx<T>*>(smem_raw); // natural order, in and out
    cpx<T>* ex = nat + CH;                             // the exchange
    const int tid = threadIdx.x, c = tid / P::NT, j = tid - c * P::NT;
    const bool lane = c < P::B;
    cpx<T>* nc = nat + (lane ? c : 0) * N;
    cpx<T>* ec = ex + (lane ? c : 0) * P::LA;
    cpx<T> tw[R1 - 1]; // stage 1 (ns = R0, k = j): w_n^(r j)
#pragma unroll
    for (int r = 1; r < R1; ++r) tw[r - 1] = lane ? wtab[r * j] : cpx<T>{(T)1, (T)0};
    const unsigned long long groups = (batch + P::B - 1) / P::B, total = batch * N;
    // plain I/O: the NEXT chunk is fetched into registers while the current one is transformed (a workgroup's loads are
    // otherwise issued and waited for in one breath, and three workgroups per CU do not cover HBM's latency: 65536 x 100 points
    // 42 -> 28 us).  (The same prefetch for the option path -- rotation and real input resolved at fetch time -- took the
    // kernels from 122-150 to 196-256 VGPRs, plain path included: its index arithmetic stays in the rolled loop below.)
    cpx<T> pre[PER];
    auto fetch = [&](unsigned long long g) {
        const unsigned long long b2 = g * CH;
        const cpx<T>* src = reinterpret_cast<const cpx<T>*>(io.in) + b2;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + q * TH;
            pre[q] = (g < groups && e < CH && b2 + e < total) ? src[e] : cpx<T>{(T)0, (T)0};
        }
    };
    if (io.plain) fetch(blockIdx.x);
    for (unsigned long long gi = blockIdx.x; gi < groups; gi += gridDim.x) {
        const unsigned long long base = gi * CH;
        __syncthreads(); // (the previous chunk's output loop is done with `nat`)
        if (io.plain) {
#pragma unroll
            for (int q = 0; q < PER; ++q)
                if (tid + q * TH < CH) nat[tid + q * TH] = pre[q];
            fetch(gi + gridDim.x);
        } else {
            for (int e = tid; e < CH; e += TH) {
                const int c2 = e / N, i0 = e - c2 * N;
                const unsigned long long v2 = gi * P::B + c2;
                unsigned i = (unsigned)i0 + io.rot_in;
                if (i >= (unsigned)N) i -= (unsigned)N;
                cpx<T> z{(T)0, (T)0};
                if (v2 < batch) {
                    if (io.in_real) z.x = io.in[v2 * N + i];
                    else z = reinterpret_cast<const cpx<T>*>(io.in)[v2 * N + i];
                }
                T w = io.in_scale;
                if (io.window_id >= 0 && !io.window_div) w = w * window_value_sym<T>(io.window_id, io.alpha, (size_t)i0, (size_t)N);
                nat[e] = cpx<T>{z.x * w, z.y * w};
            }
        }
        __syncthreads();
        if (lane && j < R1) { // stage 0 (ns = 1): v[r] = x[j + r R1]; out[j R0 + r]
            cpx<T> v[R0];
#pragma unroll
            for (int r = 0; r < R0; ++r) v[r] = nc[j + r * R1];
            mr_dft<R0, DIR>(v);
#pragma unroll
            for (int r = 0; r < R0; ++r) ec[j * P::SA + r] = v[r];
        }
        __syncthreads();
        if (lane) { // stage 1 (ns = R0, k = j < R0): v[r] = in[j + r R0] w^(r j); X[j + r R0]
            cpx<T> v[R1];
#pragma unroll
            for (int r = 0; r < R1; ++r) v[r] = ec[r * P::SA + j];
#pragma unroll
            for (int r = 1; r < R1; ++r) v[r] = twmul<DIR>(v[r], tw[r - 1]);
            mr_dft<R1, DIR>(v);
#pragma unroll
            for (int r = 0; r < R1; ++r) nc[j + r * R0] = v[r];
        }
        __syncthreads();
        if (io.plain) {
            cpx<T>* dst = reinterpret_cast<cpx<T>*>(io.out) + base;
            for (int e = tid; e < CH; e += TH)
                if (base + e < total) dst[e] = nat[e];
        } else {
            for (int e = tid; e < CH; e += TH) {
                const int c2 = e / N, k = e - c2 * N;
                const unsigned long long v2 = gi * P::B + c2;
                if (v2 >= batch) continue;
                cpx<T> z = nat[e];
                const unsigned i = (unsigned)k >= io.rot_out ? (unsigned)k - io.rot_out : (unsigned)k + (unsigned)N - io.rot_out;
                if (io.window_id >= 0 && io.window_div) {
                    const T w = window_value_sym<T>(io.window_id, io.alpha, (size_t)i, (size_t)N);
                    z = cpx<T>{z.x / w, z.y / w};
                }
                if (io.out_kind == 0) reinterpret_cast<cpx<T>*>(io.out)[v2 * N + i] = z;
                else if (io.out_kind == 1) io.out[v2 * N + i] = z.x;
                else io.out[v2 * N + i] = sizeof(T) == 4 ? (T)hypotf((float)z.x, (float)z.y) : (T)hypot((double)z.x, (double)z.y);
            }
        }
    }
}

template <typename T, int R0, int R1>
static int mr_reg2_run(const MrReg3Io<T>& io, size_t batch, bool inverse, hipStream_t s)
{
    using P = MrReg2<R0, R1, (int)sizeof(T)>;
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(P::N, &wtab));
    const size_t lds = sizeof(cpx<T>) * (size_t)P::B * (P::N + P::LA);
    static int occ = 0;
    if (occ == 0) {
        int o = 0;
        if (lds > 64 * 1024) {
            BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mr_reg2<T, -1, R0, R1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mr_reg2<T, 1, R0, R1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_mr_reg2<T, -1, R0, R1>, P::THREADS, lds) != hipSuccess || o < 1) o = 1;
        occ = o;
    }
    const size_t groups = (batch + P::B - 1) / P::B, slots = (size_t)num_cus() * (size_t)occ;
    const unsigned grid = (unsigned)(groups < slots ? groups : slots);
    if (inverse) hipLaunchKernelGGL((k_mr_reg2<T, 1, R0, R1>), dim3(grid), dim3(P::THREADS), lds, s, io, wtab, (unsigned long long)batch);
    else hipLaunchKernelGGL((k_mr_reg2<T, -1, R0, R1>), dim3(grid), dim3(P::THREADS), lds, s, io, wtab, (unsigned long long)batch);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// every n = R0 R1 < 300 that is not a power of two, radices out of k_mr_reg3's set, the smaller radix as large as possible
template <typename T>
int mr_reg2_launch(const MrReg3Io<T>& io, size_t n, size_t batch, bool inverse, hipStream_t s)
{
    static const bool off = lab_flag("BDSP_MR_NO_REG3");
    if (off) return MR_REG3_NOT_BUILT;
#define BDSP_REG2(NV, A, B_) case NV: return mr_reg2_run<T, A, B_>(io, batch, inverse, s);
    switch (n) {
    BDSP_REG2(20, 5, 4)
    BDSP_REG2(24, 6, 4)
    BDSP_REG2(25, 5, 5)
    BDSP_REG2(30, 6, 5)
    BDSP_REG2(36, 6, 6)
    BDSP_REG2(40, 8, 5)
    BDSP_REG2(45, 9, 5)
    BDSP_REG2(48, 8, 6)
    BDSP_REG2(50, 10, 5)
    BDSP_REG2(54, 9, 6)
    BDSP_REG2(60, 10, 6)
    BDSP_REG2(72, 9, 8)
    BDSP_REG2(75, 15, 5)
    BDSP_REG2(80, 10, 8)
    BDSP_REG2(81, 9, 9)
    BDSP_REG2(90, 10, 9)
    BDSP_REG2(96, 12, 8)
    BDSP_REG2(100, 10, 10)
    BDSP_REG2(108, 12, 9)
    BDSP_REG2(120, 12, 10)
    BDSP_REG2(125, 25, 5)
    BDSP_REG2(135, 15, 9)
    BDSP_REG2(144, 12, 12)
    BDSP_REG2(150, 15, 10)
    BDSP_REG2(160, 16, 10)
    BDSP_REG2(180, 15, 12)
    BDSP_REG2(192, 16, 12)
    BDSP_REG2(200, 20, 10)
    BDSP_REG2(225, 15, 15)
    BDSP_REG2(240, 16, 15)
    BDSP_REG2(250, 25, 10)
    default: break;
    }
#undef BDSP_REG2
    return MR_REG3_NOT_BUILT;
}

} // namespace bdsp
