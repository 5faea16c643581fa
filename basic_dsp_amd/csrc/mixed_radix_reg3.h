// mixed_radix_reg3.h -- the register-resident three-stage mixed-radix kernel and its launcher; included by
// mixed_radix_reg3_f32.hip and mixed_radix_reg3_f64.hip, which instantiate mr_reg3_launch<float> / <double> (two TUs: the ~60
// lengths x 2 directions compile in parallel and live in code objects of their own, which a process that never transforms
// such a batch never loads).
#pragma once
#include "bdsp_internal.h"
#include "mr_dft.h"
#include "dsp_funcs.h"

namespace bdsp {

// ---- register-resident three-stage transforms (round 6) ----------------------------------------------------------
// n = R0 R1 R2 with compile-time radices, plain I/O, large batches: the shape of the power-of-two k_fft_wg_batch.  A thread
// keeps one radix-R butterfly of each stage in registers; the data crosses LDS twice (after stage 0 and after stage 1)
// instead of four times; the workgroup is persistent over the batch, so the twiddles of stage 1 (w_{R0 R1}^{r k}, k = j mod
// R0) and stage 2 (w_n^{r j}) are per-thread constants loaded ONCE -- k_mr_wg copies the whole n-entry table into LDS for every
// two transforms and reads a twiddle per butterfly input from it.  The first stage reads HBM and the last one writes it
// straight from registers, both with unit stride across the threads of a transform.
//   Stockham indices (as in mr_stage): stage s, radix R, ns = product of the earlier radices, nb = n / R, j < nb, k = j mod ns:
//       v[r] = in[j + r nb] w_{ns R}^{r k};   out[(j / ns) ns R + k + r ns] = DFT_R(v)[r]
//   Exchange A (stage 0 -> 1): thread j writes the row j R0 + r; rows are SA = R0 | 1 apart (an odd stride in elements: the
//       64 lanes of a store hit distinct banks); stage 1 reads i = j + r R0 R2, i.e. row j / R0 + r R2, column j mod R0.
//   Exchange B (stage 1 -> 2): thread j writes (j / R0) R0 R1 + j mod R0 + r R0; groups are SB = R0 R1 + pad apart with
//       SB = R0 (mod 32), so the runs of R0 lanes tile the banks; stage 2 reads r SB + j.
//   Every LDS address is a per-thread base plus a compile-time offset.  Two buffers, two barriers per transform.
// *Measured* (tools/plan_probe.py, valid data, cold / hot us, against k_mr_wg; profiles/r06_plan_probe_valid.txt): f32 16384 x 1000
// points 136 / 115 -> 57 / 56 (0.24 -> 0.59 of the HBM roofline), 8192 x 2000 195 / 159 -> 57 / 53, 4096 x 3600 348 / 311 -> 51 / 49,
// 4096 x 3000 (512 threads) 173 / 146 -> 45 / 46, ONE 1000-point transform 7.5 / 4.8 -> 3.8 / 2.7; f64 16384 x 1000 257 / 252 -> 102 / 85,
// 8192 x 2000 498 / 489 -> 104 / 93.
template <int R0, int R1, int R2>
struct MrReg3 {
    static constexpr int N = R0 * R1 * R2;
    static constexpr int RMIN = R0 < R1 ? (R0 < R2 ? R0 : R2) : (R1 < R2 ? R1 : R2);
    static constexpr int RMAX = R0 > R1 ? (R0 > R2 ? R0 : R2) : (R1 > R2 ? R1 : R2);
    static constexpr int NT = N / RMIN;                  // threads per transform
    static constexpr int THREADS = NT <= 256 ? 256 : 512; // (3000 = 20 15 10 and four more lengths need 300 ... 400 threads)
    static constexpr int B = THREADS / NT;               // transforms per workgroup
    static constexpr int SA = R0 | 1;
    static constexpr int LA = (N / R0) * SA;
    static constexpr int SB = R0 * R1 + ((R0 - R0 * R1) % 32 + 32) % 32;
    static constexpr int LB = R2 * SB;
    static_assert(NT <= 512 && B >= 1, "a transform fits a workgroup");
};

// What the kernel fuses besides the transform: input rotation (ifft_shift), input scale, a window on the input or divided out of
// the output, real input, output rotation (fft_shift), real-part / magnitude output -- every option k_mr_wg has.  `plain` = none of them: stage 0 loads
// from HBM and stage 2 stores to it straight from registers.  Otherwise the workgroup stages its transforms' inputs and outputs
// through the two LDS buffers in natural order, in rolled loops that carry the index arithmetic (with it in the unrolled
// register code the plain path's f32 kernels went from 84 to 139 VGPRs and the f64 ones lost a wave per SIMD): three more
// barriers and two more LDS trips per transform for the calls that use an option, nothing for the ones that do not.
template <typename T>
struct MrReg3Io {
    const T* in;
    T* out;
    unsigned rot_in, rot_out; // input element i is x[(i + rot_in) mod n]; output element i is X[(i + rot_out) mod n]
    T in_scale;
    int in_real;              // the input holds n reals per vector
    int out_kind;             // 0 complex, 1 real part, 2 magnitude
    int plain;
    int window_id;            // >= 0: multiply the input by the window (evaluated like the reference, symmetrically) ...
    int window_div;           // ... or divide the output by it (windowed_ifft)
    T alpha;
};

// OPTS = false: the plain path alone.  Built for the five 512-thread lengths only: with the option code in the same kernel their f32
// instantiations need 132-150 VGPRs -- over the 128 that let TWO 8-wave workgroups share a CU -- and a plain 4096 x 3000-point
// batch went from 45 to 61 us (*measured*, rocprofv3: profiles/r06_new_kernels_kernel_stats.csv); forcing 128 registers spills.
template <typename T, int DIR, int R0, int R1, int R2, bool OPTS = true>
__global__ __launch_bounds__((MrReg3<R0, R1, R2>::THREADS)) void k_mr_reg3(MrReg3Io<T> io_, const cpx<T>* __restrict__ wtab,
                                                                            unsigned long long batch)
{
    MrReg3Io<T> io = io_;
    if constexpr (!OPTS) io.plain = 1;
    using P = MrReg3<R0, R1, R2>;
    constexpr int N = P::N, NB0 = N / R0, NB1 = N / R1, NB2 = N / R2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, c = tid / P::NT, j = tid - c * P::NT;
    const bool lane = c < P::B;
    cpx<T>* la = reinterpret_cast<cpx<T>*>(smem_raw) + (size_t)(lane ? c : 0) * (P::LA + P::LB);
    cpx<T>* lb = la + P::LA;
    const int jq = j / R0, jk = j - jq * R0;
    // per-thread twiddles, loaded once: stage 1 w_n^(r k R2), k = j mod R0; stage 2 w_n^(r j)
    cpx<T> tw1[R1 - 1], tw2[R2 - 1];
#pragma unroll
    for (int r = 1; r < R1; ++r) tw1[r - 1] = (lane && j < NB1) ? wtab[r * jk * R2] : cpx<T>{(T)1, (T)0};
#pragma unroll
    for (int r = 1; r < R2; ++r) tw2[r - 1] = (lane && j < NB2) ? wtab[r * j] : cpx<T>{(T)1, (T)0};
    const unsigned long long groups = (batch + P::B - 1) / P::B;
    cpx<T>* const lds0 = reinterpret_cast<cpx<T>*>(smem_raw);
    for (unsigned long long gi = blockIdx.x; gi < groups; gi += gridDim.x) {
        const unsigned long long vec = gi * P::B + c;
        const bool active = lane && vec < batch;
        if (!io.plain) {
            // fused input options: the workgroup brings its transforms' inputs into the B buffers in natural order (rotation,
            // real input, scale applied on the way), so that stage 0 below finds them at compile-time offsets -- the index
            // arithmetic lives in this rolled loop, not in the R0 registers of every thread
            __syncthreads(); // (the previous iteration's stage 2 / output loop is done with both buffers)
            for (int e = tid; e < N * P::B; e += P::THREADS) {
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
                lds0[(size_t)c2 * (P::LA + P::LB) + P::LA + i0] = cpx<T>{z.x * w, z.y * w};
            }
            __syncthreads();
        }
        if (lane && j < NB0) {
            cpx<T> v[R0];
            if (io.plain) {
                const cpx<T>* src = reinterpret_cast<const cpx<T>*>(io.in) + vec * N + j;
#pragma unroll
                for (int r = 0; r < R0; ++r) v[r] = active ? src[r * NB0] : cpx<T>{(T)0, (T)0};
            } else {
#pragma unroll
                for (int r = 0; r < R0; ++r) v[r] = lb[j + r * NB0];
            }
            mr_dft<R0, DIR>(v);
#pragma unroll
            for (int r = 0; r < R0; ++r) la[j * P::SA + r] = v[r];
        }
        __syncthreads();
        if (lane && j < NB1) {
            cpx<T> v[R1];
#pragma unroll
            for (int r = 0; r < R1; ++r) v[r] = la[(jq + r * R2) * P::SA + jk];
#pragma unroll
            for (int r = 1; r < R1; ++r) v[r] = twmul<DIR>(v[r], tw1[r - 1]);
            mr_dft<R1, DIR>(v);
#pragma unroll
            for (int r = 0; r < R1; ++r) lb[jq * P::SB + jk + r * R0] = v[r];
        }
        __syncthreads();
        if (lane && j < NB2) {
            cpx<T> v[R2];
#pragma unroll
            for (int r = 0; r < R2; ++r) v[r] = lb[r * P::SB + j];
#pragma unroll
            for (int r = 1; r < R2; ++r) v[r] = twmul<DIR>(v[r], tw2[r - 1]);
            mr_dft<R2, DIR>(v);
            if (io.plain) {
                if (active) {
                    cpx<T>* dst = reinterpret_cast<cpx<T>*>(io.out) + vec * N + j;
#pragma unroll
                    for (int r = 0; r < R2; ++r) dst[r * NB2] = v[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < R2; ++r) la[j + r * NB2] = v[r]; // (everybody read the A buffer before the last barrier)
            }
        }
        if (!io.plain) {
            // fused output options, the same way: the spectrum in natural order out of the A buffers
            __syncthreads();
            for (int e = tid; e < N * P::B; e += P::THREADS) {
                const int c2 = e / N, k = e - c2 * N;
                const unsigned long long v2 = gi * P::B + c2;
                if (v2 >= batch) continue;
                cpx<T> z = lds0[(size_t)c2 * (P::LA + P::LB) + k];
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

constexpr int MR_REG3_NOT_BUILT = 1 << 20;

template <typename T, int R0, int R1, int R2>
static int mr_reg3_run(const MrReg3Io<T>& io, size_t batch, bool inverse, hipStream_t s)
{
    using P = MrReg3<R0, R1, R2>;
    const cpx<T>* wtab;
    BDSP_TRY(twiddle_table<T>(P::N, &wtab));
    const size_t lds = sizeof(cpx<T>) * (size_t)P::B * (P::LA + P::LB);
    static int occ = 0; // resident workgroups per CU as the runtime computes it (registers and LDS), once per instantiation
    if (occ == 0) {
        int o = 0;
        if (lds > 64 * 1024) {
            BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mr_reg3<T, -1, R0, R1, R2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mr_reg3<T, 1, R0, R1, R2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_mr_reg3<T, -1, R0, R1, R2>, P::THREADS, lds) != hipSuccess || o < 1) o = 1;
        occ = o;
    }
    const size_t groups = (batch + P::B - 1) / P::B, slots = (size_t)num_cus() * (size_t)occ;
    // (LAB: BDSP_MR_REG3_ROUNDS = r sends batches of fewer than r rounds of persistent workgroups to k_mr_wg instead)
    static const int min_rounds = [] { const char* e = lab_env("BDSP_MR_REG3_ROUNDS"); return e ? atoi(e) : 0; }();
    if (groups < (size_t)min_rounds * slots) return MR_REG3_NOT_BUILT;
    const unsigned grid = (unsigned)(groups < slots ? groups : slots);
    if constexpr (P::THREADS == 512 && sizeof(T) == 4) {
        if (io.plain) { // the plain-only instantiation: two workgroups per CU (see k_mr_reg3 OPTS)
            static int occ_plain = 0;
            if (occ_plain == 0) {
                int o = 0;
                if (lds > 64 * 1024) {
                    BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mr_reg3<T, -1, R0, R1, R2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mr_reg3<T, 1, R0, R1, R2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                }
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_mr_reg3<T, -1, R0, R1, R2, false>, P::THREADS, lds) != hipSuccess || o < 1) o = 1;
                occ_plain = o;
            }
            const size_t slots_p = (size_t)num_cus() * (size_t)occ_plain;
            const unsigned grid_p = (unsigned)(groups < slots_p ? groups : slots_p);
            if (inverse) hipLaunchKernelGGL((k_mr_reg3<T, 1, R0, R1, R2, false>), dim3(grid_p), dim3(P::THREADS), lds, s, io, wtab, (unsigned long long)batch);
            else hipLaunchKernelGGL((k_mr_reg3<T, -1, R0, R1, R2, false>), dim3(grid_p), dim3(P::THREADS), lds, s, io, wtab, (unsigned long long)batch);
            BDSP_LAUNCH_CHECK();
            return BDSP_OK;
        }
    }
    if (inverse) hipLaunchKernelGGL((k_mr_reg3<T, 1, R0, R1, R2>), dim3(grid), dim3(P::THREADS), lds, s, io, wtab, (unsigned long long)batch);
    else hipLaunchKernelGGL((k_mr_reg3<T, -1, R0, R1, R2>), dim3(grid), dim3(P::THREADS), lds, s, io, wtab, (unsigned long long)batch);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// The lengths: every n = R0 R1 R2 <= 4096 (f64: <= 2048, like k_mr_wg) that is not a power of two, with radices out of
// {4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25}, a factorisation whose n / min radix threads fit a 256-thread workgroup where one exists
// (else 512 threads: five lengths), the smallest radix as large as possible, the largest radix first (stage 0 needs no twiddle
// registers).  tools/gen_reg3_table.py prints this table.
template <typename T>
int mr_reg3_launch(const MrReg3Io<T>& io, size_t n, size_t batch, bool inverse, hipStream_t s)
{
    static const bool off = lab_flag("BDSP_MR_NO_REG3");
    if (off) return MR_REG3_NOT_BUILT;
#define BDSP_REG3(NV, A, B_, C_) case NV: return mr_reg3_run<T, A, B_, C_>(io, batch, inverse, s);
    switch (n) {
    BDSP_REG3(300, 10, 6, 5)
    BDSP_REG3(320, 8, 8, 5)
    BDSP_REG3(324, 9, 6, 6)
    BDSP_REG3(360, 10, 6, 6)
    BDSP_REG3(375, 15, 5, 5)
    BDSP_REG3(384, 8, 8, 6)
    BDSP_REG3(400, 10, 8, 5)
    BDSP_REG3(405, 9, 9, 5)
    BDSP_REG3(432, 9, 8, 6)
    BDSP_REG3(450, 10, 9, 5)
    BDSP_REG3(480, 10, 8, 6)
    BDSP_REG3(486, 9, 9, 6)
    BDSP_REG3(500, 10, 10, 5)
    BDSP_REG3(540, 10, 9, 6)
    BDSP_REG3(576, 9, 8, 8)
    BDSP_REG3(600, 10, 10, 6)
    BDSP_REG3(625, 25, 5, 5)
    BDSP_REG3(640, 10, 8, 8)
    BDSP_REG3(648, 9, 9, 8)
    BDSP_REG3(675, 15, 9, 5)
    BDSP_REG3(720, 10, 9, 8)
    BDSP_REG3(729, 9, 9, 9)
    BDSP_REG3(750, 15, 10, 5)
    BDSP_REG3(768, 12, 8, 8)
    BDSP_REG3(800, 10, 10, 8)
    BDSP_REG3(810, 10, 9, 9)
    BDSP_REG3(864, 12, 9, 8)
    BDSP_REG3(900, 10, 10, 9)
    BDSP_REG3(960, 12, 10, 8)
    BDSP_REG3(972, 12, 9, 9)
    BDSP_REG3(1000, 10, 10, 10)
    BDSP_REG3(1080, 12, 10, 9)
    BDSP_REG3(1125, 15, 15, 5)
    BDSP_REG3(1152, 12, 12, 8)
    BDSP_REG3(1200, 12, 10, 10)
    BDSP_REG3(1215, 15, 9, 9)
    BDSP_REG3(1250, 25, 10, 5)
    BDSP_REG3(1280, 16, 10, 8)
    BDSP_REG3(1296, 12, 12, 9)
    BDSP_REG3(1350, 15, 10, 9)
    BDSP_REG3(1440, 12, 12, 10)
    BDSP_REG3(1500, 15, 10, 10)
    BDSP_REG3(1536, 16, 12, 8)
    BDSP_REG3(1600, 16, 10, 10)
    BDSP_REG3(1620, 15, 12, 9)
    BDSP_REG3(1728, 12, 12, 12)
    BDSP_REG3(1800, 15, 12, 10)
    BDSP_REG3(1875, 25, 15, 5)
    BDSP_REG3(1920, 16, 12, 10)
    BDSP_REG3(2000, 20, 10, 10)
    BDSP_REG3(2025, 15, 15, 9)
    default: break;
    }
    if constexpr (sizeof(T) == 4) {
        switch (n) {
        BDSP_REG3(2160, 15, 12, 12)
        BDSP_REG3(2250, 15, 15, 10)
        BDSP_REG3(2304, 16, 12, 12)
        BDSP_REG3(2400, 16, 15, 10)
        BDSP_REG3(2500, 25, 10, 10)
        BDSP_REG3(2560, 16, 16, 10)
        BDSP_REG3(2700, 15, 15, 12)
        BDSP_REG3(2880, 16, 15, 12)
        BDSP_REG3(3000, 20, 15, 10)
        BDSP_REG3(3072, 16, 16, 12)
        BDSP_REG3(3200, 20, 16, 10)
        BDSP_REG3(3375, 15, 15, 15)
        BDSP_REG3(3600, 16, 15, 15)
        BDSP_REG3(3750, 25, 15, 10)
        BDSP_REG3(3840, 16, 16, 15)
        BDSP_REG3(4000, 20, 20, 10)
        default: break;
        }
    }
#undef BDSP_REG3
    return MR_REG3_NOT_BUILT;
}

} // namespace bdsp
