// interp.hip -- InterpolationOps::interpolatef (vector/src/vector_types/time_freq/interpolation.rs:387-482).
//
// Polyphase resampling with wrap-around.  The reference has two code paths with slightly different
// tap windows and this backend follows the same dispatch so the
// results agree with whichever path the CPU would have taken:
//   scalar path  (interpolate_priv_scalar :92-131): any factor; taps evaluated per output
//       y[i] = sum_{m=0}^{2L} x[(r - L + m) mod N] * f(-L - (t - r) + d + m),  t = i/factor, r = floor(t)
//   "simd" path  (interpolate_priv_simd :191-290): integer factor, L <= 202, new_len >= 2000;
//       per-phase tap vectors taps_s[m] = f(-(L-1) + d + m - s/factor) (function_to_vectors :133-181);
//       edges (first/last (2L+1)*factor outputs, interpolate_priv_simd_step :293-315):
//           y[i] = sum_m x[(r - L + 1 + m) mod N] * taps_{i mod f}[m],          r = i div f
//       inner region (register dot product :249-273):
//           y[i] = sum_m x[c + L - 1 - m] * taps_{(f - i mod f) mod f}[m],      c = ceil(i/f)
// All tap arguments are accumulated in T by repeated +1 exactly as the reference does.
#include "bdsp_internal.h"
#include <cstring>
#include <map>
#include <mutex>
#include "dsp_funcs.h"
#include <vector>

namespace bdsp {

template <typename T> __device__ __forceinline__ T dev_floor(T x);
template <> __device__ __forceinline__ float dev_floor<float>(float x) { return floorf(x); }
template <> __device__ __forceinline__ double dev_floor<double>(double x) { return floor(x); }

template <typename T> __device__ __forceinline__ T dev_fma(T a, T b, T c);
template <> __device__ __forceinline__ float dev_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double dev_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T>
size_t interpolatef_new_len(size_t len, T factor)
{
    // interpolation.rs:406-410: round(len * factor) in T, made even
    T v = (T)len * factor;
    size_t new_len = (size_t)(sizeof(T) == 4 ? roundf((float)v) : round((double)v));
    return new_len + new_len % 2;
}
template size_t interpolatef_new_len<float>(size_t, float);
template size_t interpolatef_new_len<double>(size_t, double);

// taps[s*ntaps + m] = f(-(L-1) + delay + m - s/factor)
template <typename T>
__global__ void k_interp_taps(T* __restrict__ taps, int fid, T rolloff, int conv_len, int factor, T delay)
{
    // one thread per tap; each repeats the reference's own accumulation j = j + 1 up to its index, so the
    // arguments are bit-identical to the sequential loop (the serial version took 9-14 us for 100 taps)
    const int ntaps = 2 * conv_len + 1;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= factor * ntaps) return;
    const int s = id / ntaps, m = id % ntaps;
    T offset = (T)s / (T)factor;
    T j = -((T)conv_len - (T)1) + delay;
    for (int k = 0; k < m; ++k) j = j + (T)1;
    taps[id] = conv_time_value<T>(fid, rolloff, j - offset);
}

// The table path for the outputs outside [skip_lo, skip_hi) (all of them when the range is empty): workgroup `block` of
// `nblocks`.  lt: LDS room for the ntaps * factor taps.
template <typename T, bool CPLX>
__device__ __forceinline__ void interp_table_body(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ taps, T* lt,
                                                  long long points, long long new_points, int conv_len, int factor,
                                                  long long skip_lo, long long skip_hi, unsigned block, unsigned nblocks)
{
    const int ntaps = 2 * conv_len + 1;
    for (int k = threadIdx.x; k < ntaps * factor; k += blockDim.x) lt[k] = taps[k];
    __syncthreads();
    const long long scalar_len = (long long)ntaps * factor;
    // outputs in [skip_lo, skip_hi) belong to the blocked inner kernel: walk only the two edge runs
    const long long nskip = skip_hi > skip_lo ? skip_hi - skip_lo : 0;
    for (long long g = (long long)block * blockDim.x + threadIdx.x; g < new_points - nskip;
         g += (long long)nblocks * blockDim.x) {
        const long long i = (nskip && g >= skip_lo) ? g + nskip : g;
        T sr = 0, si = 0;
        const bool edge = i < scalar_len || i + scalar_len >= new_points || new_points < 2 * scalar_len;
        if (edge) {
            long long r = i / factor;
            const T* t = lt + (i % factor) * ntaps;
            long long pos = (r - conv_len) % points;
            if (pos < 0) pos += points;
            for (int m = 0; m < ntaps; ++m) {
                pos = pos + 1 < points ? pos + 1 : 0;
                if (CPLX) {
                    T re = x[2 * pos], im = x[2 * pos + 1];
                    sr = sr + (re * t[m] - im * (T)0);
                    si = si + (re * (T)0 + im * t[m]);
                } else sr = sr + x[pos] * t[m];
            }
        } else {
            long long end = (i + factor - 1) / factor + conv_len;
            int shift = (int)((factor - i % factor) % factor);
            const T* t = lt + shift * ntaps;
            for (int m = ntaps - 1; m >= 0; --m) {
                long long n = end - 1 - m;
                if (CPLX) {
                    sr = sr + x[2 * n] * t[m];
                    si = si + x[2 * n + 1] * t[m];
                } else sr = sr + x[n] * t[m];
            }
        }
        if (CPLX) { y[2 * i] = sr; y[2 * i + 1] = si; }
        else y[i] = sr;
    }
}


template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_interp_table(const T* __restrict__ x, T* __restrict__ y,
                                                       const T* __restrict__ taps, long long points,
                                                       long long new_points, int conv_len, int factor,
                                                       long long skip_lo, long long skip_hi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    interp_table_body<T, CPLX>(x, y, taps, reinterpret_cast<T*>(smem_raw), points, new_points, conv_len, factor, skip_lo, skip_hi,
                               blockIdx.x, gridDim.x);
}

// Inner region of the integer-factor ("simd") path, register/LDS blocked: a thread owns one input
// position q and produces the FACTOR outputs i = FACTOR*q + s.  With c = ceil(i/f) the reference sums
//   s = 0 :  sum_m x[q + L - 1 - m] * taps_0[m]          s > 0 :  sum_m x[q + L - m] * taps_{f-s}[m]
// (interpolation.rs:249-273), lowest address first.  Walking n = q-L-1 .. q+L once, x[n] feeds output
// s = 0 with tap m = q+L-1-n and outputs s > 0 with tap m = q+L-n, so each input sample is read from
// LDS once for FACTOR outputs (the one-output-per-thread kernel re-reads it FACTOR times through L1
// and measured 14 % of the HBM roofline on config C4).  Results cross threads through LDS so the
// stores are contiguous.  The workgroup covers q in [q0, q0+256) of the inner region
// [q_lo, q_hi) = positions whose FACTOR outputs are all inner outputs.
// QB consecutive positions per thread: a tap read from LDS (a broadcast, but still an LDS instruction) then
// feeds QB * FACTOR multiply-adds instead of FACTOR -- with one position per thread the kernel was bound
// by LDS instruction issue (39 LDS reads per output point; *measured* 98 us for config C4b).
// Round 3: the two edge runs (the few hundred outputs next to the vector's ends, which take the table path with its
// wrap-around) ride in the SAME launch: the workgroups past `inner_blocks` run interp_table_body on them.  As a separate
// launch in front of this one they cost 4.8 us of config C4b's 88.
template <typename T, bool CPLX, int FACTOR, int QB>
__global__ __launch_bounds__(256) void k_interp_inner(const T* __restrict__ x, T* __restrict__ y,
                                                      const T* __restrict__ taps, long long q_lo,
                                                      long long q_hi, int conv_len, long long points,
                                                      long long new_points, unsigned inner_blocks, int stream_out)
{
    constexpr int E = CPLX ? 2 : 1;
    constexpr int TILE = 256 * QB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (blockIdx.x >= inner_blocks) {
        interp_table_body<T, CPLX>(x, y, taps, reinterpret_cast<T*>(smem_raw), points, new_points, conv_len, FACTOR,
                                   q_lo * FACTOR, q_hi * FACTOR, blockIdx.x - inner_blocks, gridDim.x - inner_blocks);
        return;
    }
    const int ntaps = 2 * conv_len + 1;
    T* lt = reinterpret_cast<T*>(smem_raw);                 // [ntaps + 1][FACTOR]
    T* lx = lt + ((FACTOR * (ntaps + 1) + 3) & ~3);         // [TILE + 2L + 2][E]
    T* lo = lx + ((TILE + 2 * conv_len + 2) * E + 3 & ~3);  // [TILE * FACTOR][E] output staging
    const int t = threadIdx.x;
    const long long q0 = q_lo + (long long)blockIdx.x * TILE;
    // tap table re-laid out per step j of the walk below: lt[j*FACTOR + s] is the tap that x[q-L-1+j] carries
    // into output s (0 where it carries none), so one wide LDS read per step fetches all FACTOR taps:
    //   s = 0: m = 2L - j (j <= 2L)          s > 0: m = 2L + 1 - j (j >= 1), tap vector f - s
    for (int k = t; k < (ntaps + 1) * FACTOR; k += 256) {
        const int j = k / FACTOR, sidx = k % FACTOR;
        T w = (T)0;
        if (sidx == 0) { if (j <= 2 * conv_len) w = taps[2 * conv_len - j]; }
        else if (j >= 1) w = taps[(FACTOR - sidx) * ntaps + 2 * conv_len + 1 - j];
        lt[k] = w;
    }
    // x[q0 - L - 1 .. q0 + TILE - 1 + L]  (always in range below: the inner region starts (2L+1) positions in)
    const int span = TILE + 2 * conv_len + 2;
    const long long xbase = q0 - conv_len - 1;
    for (int k = t; k < span * E; k += 256) {
        long long g = xbase * E + k;
        lx[k] = g < points * E ? x[g] : (T)0; // the last workgroup's tile may overhang the vector
    }
    __syncthreads();
    T ar[QB][FACTOR], ai[QB][FACTOR];
#pragma unroll
    for (int k = 0; k < QB; ++k)
#pragma unroll
        for (int s = 0; s < FACTOR; ++s) { ar[k][s] = 0; ai[k][s] = 0; }
    // position q = q0 + QB*t + k; local index of x[n] in lx: n - xbase = QB*t + k + j, j = 0 .. 2L+1
    if constexpr (QB == 1) {
        // one position per thread: step j needs just x[t + j] -- one LDS read per step, no sliding window to shuffle
        // (the generic loop below spent 8 of its 24 vector instructions per two steps on register moves and two
        // branches on its conditional last read).  2L + 2 steps: always an even count.
#pragma unroll 2
        for (int j = 0; j <= 2 * conv_len + 1; ++j) {
            const T xr = lx[(t + j) * E];
            const T xi = CPLX ? lx[(t + j) * E + 1] : (T)0;
            T w[FACTOR];
#pragma unroll
            for (int s = 0; s < FACTOR; ++s) w[s] = lt[j * FACTOR + s];
#pragma unroll
            for (int s = 0; s < FACTOR; ++s) {
                ar[0][s] = dev_fma<T>(xr, w[s], ar[0][s]);
                if (CPLX) ai[0][s] = dev_fma<T>(xi, w[s], ai[0][s]);
            }
        }
    } else {
    T wr[QB], wi[QB]; // sliding window x[QB*t + k + j], k = 0..QB-1
#pragma unroll
    for (int k = 0; k < QB; ++k) {
        wr[k] = lx[(QB * t + k) * E];
        wi[k] = CPLX ? lx[(QB * t + k) * E + 1] : (T)0;
    }
#pragma unroll 2
    for (int j = 0; j <= 2 * conv_len + 1; ++j) {
        T w[FACTOR];
#pragma unroll
        for (int s = 0; s < FACTOR; ++s) w[s] = lt[j * FACTOR + s];
#pragma unroll
        for (int s = 0; s < FACTOR; ++s)
#pragma unroll
            for (int k = 0; k < QB; ++k) {
                // fused multiply-add: one rounding instead of the reference's two per tap (the result is compared with
                // tolerance) and half the VALU work of this VALU-heavy kernel
                ar[k][s] = dev_fma<T>(wr[k], w[s], ar[k][s]);
                if (CPLX) ai[k][s] = dev_fma<T>(wi[k], w[s], ai[k][s]);
            }
        // slide the window by one sample
#pragma unroll
        for (int k = 0; k + 1 < QB; ++k) { wr[k] = wr[k + 1]; wi[k] = wi[k + 1]; }
        const int nx = (QB * t + QB + j) * E; // < span*E for every j <= 2L+1 except the last step's read
        if (j <= 2 * conv_len) {
            wr[QB - 1] = lx[nx];
            wi[QB - 1] = CPLX ? lx[nx + 1] : (T)0;
        }
    }
    }
    long long nq = q_hi - q0;
    if (nq > TILE) nq = TILE;
    const long long out0 = q0 * FACTOR * E, nout = nq * FACTOR * E;
    constexpr int VN = 16 / sizeof(T); // scalars per 16-byte packet
    constexpr int PER_THREAD = QB * FACTOR * E; // scalars a thread hands over
    if constexpr ((FACTOR * E) % VN == 0) {
        // Every POSITION contributes whole 16-byte packets (so out0, a multiple of FACTOR*E scalars, is 16-byte aligned)
        // and every thread PP of them, contiguous in the output.  Written as they lie, the PP
        // ds_write_b128 of a thread start PP*16 bytes after its neighbour's: the hardware serves such a store in groups
        // of 8 lanes over 32 dword banks, so with PP = 4 lanes 0/2/4/6 meet on one bank -- four-way conflicts, the 20 %
        // LDS conflict cycles the round-2 PMC showed for config C4b.  The packets of thread t are therefore ROTATED
        // within its own PP slots by (t / (8 / PP)) mod PP (PP < 8) or t mod PP: eight consecutive lanes then cover
        // eight different 16-byte bank groups, and the reader -- which takes packet k = PP*t' + part from slot
        // PP*t' + ((part + rot(t')) mod PP) -- still sees each aligned run of PP slots exactly once per PP lanes.
        constexpr int PP = PER_THREAD / VN;
        typedef T vecT __attribute__((ext_vector_type(VN)));
        vecT* lov = reinterpret_cast<vecT*>(lo);
        auto rot = [](int q) { return PP >= 8 ? (q & (PP - 1)) : (PP > 1 ? ((q / (8 / (PP > 1 ? PP : 1))) & (PP - 1)) : 0); };
        static_assert((PP & (PP - 1)) == 0 || PP == 3 || PP == 6 || PP == 12, "rotation below handles any PP");
        T flat[PER_THREAD];
#pragma unroll
        for (int k = 0; k < QB; ++k)
#pragma unroll
            for (int s = 0; s < FACTOR; ++s) {
                flat[(k * FACTOR + s) * E] = ar[k][s];
                if (CPLX) flat[(k * FACTOR + s) * E + 1] = ai[k][s];
            }
        constexpr bool POW2 = (PP & (PP - 1)) == 0;
        const int r = POW2 ? rot(t) : 0;
#pragma unroll
        for (int part = 0; part < PP; ++part) {
            vecT pk;
#pragma unroll
            for (int e = 0; e < VN; ++e) pk[e] = flat[part * VN + e];
            lov[PP * t + (POW2 ? ((part + r) & (PP - 1)) : part)] = pk;
        }
        __syncthreads();
        vecT* yv = reinterpret_cast<vecT*>(y + out0);
        // A result too large for the 256 MB Infinity Cache to hold until its reader comes is STREAMED (non-temporal
        // stores): config C4b's 256 MB ran 82 -> 69 us that way; a result that fits (128 MB: 46 -> 48 us) is not.
        if (stream_out) {
            for (long long k = t; k < nout / VN; k += 256) {
                const int q = (int)(k / PP), part = (int)(k % PP);
                __builtin_nontemporal_store(lov[PP * q + (POW2 ? ((part + rot(q)) & (PP - 1)) : part)], &yv[k]);
            }
        } else {
            for (long long k = t; k < nout / VN; k += 256) {
                const int q = (int)(k / PP), part = (int)(k % PP);
                yv[k] = lov[PP * q + (POW2 ? ((part + rot(q)) & (PP - 1)) : part)];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < QB; ++k)
#pragma unroll
            for (int s = 0; s < FACTOR; ++s) {
                lo[((QB * t + k) * FACTOR + s) * E] = ar[k][s];
                if (CPLX) lo[((QB * t + k) * FACTOR + s) * E + 1] = ai[k][s];
            }
        __syncthreads();
        for (long long k = t; k < nout; k += 256) y[out0 + k] = lo[k];
    }
}

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_interp_scalar(const T* __restrict__ x, T* __restrict__ y,
                                                        long long points, long long new_points,
                                                        int conv_len, T factor, T delay, int fid, T rolloff)
{
    double rot_s = 0.0, rot_c = 1.0;
    if (fid != 0) sincospi((double)rolloff, &rot_s, &rot_c);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < new_points;
         i += (long long)gridDim.x * blockDim.x) {
        T center = (T)i / factor;
        T rounded = dev_floor<T>(center);
        long long pos = ((long long)rounded - conv_len - 1) % points;
        if (pos < 0) pos += points;
        T j = -(T)conv_len - (center - rounded) + delay;
        T sr = 0, si = 0;
        // The tap arguments of one output are j, j+1, j+2, ...: sin(pi (j+k)) = (-1)^k sin(pi j), and the raised
        // cosine's cos(pi beta (j+k)) follows by rotating (cos, sin)(pi beta j) by pi beta per tap -- ONE sincospi
        // pair per output instead of a sin and a cos per tap (the per-tap version spent 0.4-1.5 ms on
        // 4M -> 10M points).  Trigonometry in double, everything else in T in the reference's order
        // (conv_types.rs:406-424); the removable singularities are tested on the same accumulated j.
        const T one = (T)1, two = (T)2, pi = (T)3.14159265358979323846;
        double sj = sinpi((double)j), cbj = 1.0, sbj = 0.0;
        if (fid != 0) sincospi((double)rolloff * (double)j, &sbj, &cbj);
        for (int k = 0; k < 2 * conv_len + 1; ++k) {
            pos = pos + 1 < points ? pos + 1 : 0;
            T w;
            if (j == (T)0) w = one;
            else if (fid == 0) {
                T pi_x = pi * j;
                w = (T)sj / pi_x;
            } else if (dev_abs(j) == one / (two * rolloff)) {
                T arg = pi / two / rolloff;
                w = dev_sin(arg) / arg * pi / (two * two);
            } else {
                T pi_x = pi * j;
                T arg = two * rolloff * j;
                const T t = one - dev_abs(arg);
                if (dev_abs(t) < (T)0.25) w = (T)sj * rc_near_num<T>(t) / pi_x / (two - t); // (dsp_funcs.h: no cancellation)
                else w = (T)sj * (T)cbj / pi_x / (one - (arg * arg));
            }
            sj = -sj;
            if (fid != 0) {
                const double c = cbj * rot_c - sbj * rot_s;
                sbj = sbj * rot_c + cbj * rot_s;
                cbj = c;
            }
            if (CPLX) {
                T re = x[2 * pos], im = x[2 * pos + 1];
                sr = sr + (re * w - im * (T)0);
                si = si + (re * (T)0 + im * w);
            } else sr = sr + x[pos] * w;
            j = j + (T)1;
        }
        if (CPLX) { y[2 * i] = sr; y[2 * i + 1] = si; }
        else y[i] = sr;
    }
}

// Round 4: the scalar path again, two to three times faster (fractional factors are what interpolatef is FOR: 44.1 -> 48 kHz
// is a factor of 1.088).  Same arithmetic as k_interp_scalar where the reference's rounding matters -- j accumulated in
// T, the sum taken tap by tap in the reference's order with the (w, 0) complex product spelled out -- but per tap
//   * the raised cosine's cos(pi beta (j0 + k)) = cb0 C_k - sb0 S_k from a per-launch LDS table of cos / sin(pi beta k)
//     (k = 0 .. 2L, built in double) instead of a four-multiply-add rotation in double carried from tap to tap,
//   * ONE division: sj c / (pi j (1 - (2 beta j)^2)) (the reference divides twice; the quotient differs by an ulp),
//   * the two removable singularities by selects instead of branches, the vector read as one 8- or 16-byte load.
// *Measured* (tools/interp_frac_bench.py, 4M -> 10M points, factor 2.5, conv_len 12): see DESIGN.md 4.4.
template <typename T> __device__ __forceinline__ T quot(T a, T b) { return a / b; }
// f32: the quotient through v_rcp_f32 (1 ulp) -- a correctly rounded division is a dozen instructions on this unit, a
// third of the tap (the kernel is bound by its instruction count: ~25 taps x 10M outputs at 4-5 clocks per wave
// instruction); the weights move by an ulp or two, three orders below the 1e-6 the path is held to
template <> __device__ __forceinline__ float quot<float>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_interp_scalar_v2(const T* __restrict__ x, T* __restrict__ y,
                                                           long long points_, long long new_points,
                                                           int conv_len, T factor, T delay, int fid, T rolloff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tabc = reinterpret_cast<T*>(smem_raw);
    const int ntaps = 2 * conv_len + 1;
    T* tabs = tabc + ntaps;
    if (fid != 0) {
        for (int k = threadIdx.x; k < ntaps; k += 256) {
            double sk, ck;
            sincospi((double)rolloff * (double)k, &sk, &ck);
            tabc[k] = (T)ck;
            tabs[k] = (T)sk;
        }
        __syncthreads();
    }
    typedef T vec2 __attribute__((ext_vector_type(2)));
    typedef int IDX; // 32-bit positions in the tap loop: the launcher sends vectors of 2^31 points or more to k_interp_scalar
    const IDX points = (IDX)points_;
    const T one = (T)1, two = (T)2, pi = (T)3.14159265358979323846;
    // the value at the raised cosine's second singularity, |j| = 1 / (2 beta) (conv_types.rs:406-424)
    T wsing = one;
    const T jsing = fid != 0 ? one / (two * rolloff) : (T)-1;
    if (fid != 0) {
        const T arg = pi / two / rolloff;
        wsing = dev_sin(arg) / arg * pi / (two * two);
    }
    const T tworo = two * rolloff;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < new_points;
         i += (long long)gridDim.x * blockDim.x) {
        const T center = (T)i / factor;
        const T rounded = dev_floor<T>(center);
        long long p0 = ((long long)rounded - conv_len - 1) % points_;
        if (p0 < 0) p0 += points_;
        IDX pos = (IDX)p0;
        T j = -(T)conv_len - (center - rounded) + delay;
        // (sinpi / sincospi reduce their argument exactly, so T's own versions are as good as the double ones here; the
        // product beta * j rounds like the reference's own pi * x * beta does)
        T sj, cdummy;
        dev_sincospi<T>(j, &sj, &cdummy);
        T cb0 = one, sb0 = (T)0;
        if (fid != 0) dev_sincospi<T>(rolloff * j, &sb0, &cb0);
        vec2 acc = vec2{(T)0, (T)0};
        T sr = 0;
#pragma unroll 2
        for (int k = 0; k < ntaps; ++k) {
            pos = pos + 1 < points ? pos + 1 : 0;
            const T pi_x = pi * j;
            T w;
            if (fid == 0) {
                w = quot<T>(sj, pi_x);
            } else {
                const T arg = tworo * j;
                const T t = one - dev_abs(arg);
                if (dev_abs(t) < (T)0.25) {
                    // near the second singularity, |2 beta j| -> 1: the cancellation-free form of dsp_funcs.h (rc_near_num).
                    // (Round 5: the table product's 1e-7 of absolute error on a cosine that is itself only 1e-2 put the f32
                    // result 3.6e-6 from the oracle -- four times the reference's own rounding -- for roll-off 0.35, conv_len
                    // 20; with this branch 9e-7, the oracle's own distance from the exact weights.  Two to four of the 2 L + 1
                    // tap iterations of a wave take it.  Skipping the test for the tap indices that cannot come near -- a uniform
                    // range per launch -- measured SLOWER: raised cosine f32 288 -> 323 us, f64 471 -> 492.)
                    w = quot<T>(sj * rc_near_num<T>(t), pi_x * (two - t));
                } else {
                    const T c = cb0 * tabc[k] - sb0 * tabs[k];
                    w = quot<T>(sj * c, pi_x * (one - arg * arg));
                }
                w = dev_abs(j) == jsing ? wsing : w;
            }
            w = j == (T)0 ? one : w;
            sj = -sj;
            if (CPLX) {
                // (re, im) x (w, 0) spelled out like the reference's complex product, as two-wide operations
                const vec2 z = reinterpret_cast<const vec2*>(x)[pos];
                const vec2 zw = z * vec2{w, w}, z0 = vec2{z.y, z.x} * vec2{(T)0, (T)0};
                acc = acc + vec2{zw.x - z0.x, zw.y + z0.y};
            } else sr = sr + x[pos] * w;
            j = j + (T)1;
        }
        if (CPLX) reinterpret_cast<vec2*>(y)[i] = acc;
        else y[i] = sr;
    }
}

// Round 6: the fractional path with TWO TAPS PER PACKED INSTRUCTION (f32: v_pk_mul_f32 / v_pk_add_f32 on (tap k, tap k+1); f64
// has no packed arithmetic and takes the same structure on pairs of scalar operations -- what it gains is everything else).
// k_interp_scalar_v2 is bound by its instruction count -- ~35 wave instructions per tap in its ISA: position wrap (3) and
// 64-bit address arithmetic (2) per tap, two LDS reads with an immediate wait, the near-singularity test and its divergent
// branch, two selects, the (w, 0) complex product spelled out.  Here
//   * a wave whose 2 L + 1 taps do not cross the end of the vector (all but the first and last few outputs) reads x through
//     ONE base pointer with immediate offsets -- no wrap test, no address arithmetic per tap;
//   * the weights of a pair of taps come from packed arithmetic on (j, j+1): pi j, 2 beta j, its square, 1 - that, the
//     denominator product, cb0 C_k - sb0 S_k with the table read as 8-byte pairs, the numerator with the sign pattern
//     (sj, -sj), the final product -- 11 packed instructions and two v_rcp_f32 for two taps;
//   * the near-singularity polynomial, the select at the singularity and the select at j == 0 can only apply to a handful of
//     tap indices that are the SAME for every output of a launch (j = k - L - frac + delay, frac in [0, 1)): the host marks
//     those pairs in `slow_pairs` and only they run the per-tap code of k_interp_scalar_v2 -- a wave-uniform branch on a
//     scalar bit, no divergence, nothing tested per tap elsewhere;
//   * the complex product is z * (w, w): the reference's "- im * 0" and "+ re * 0" terms are +-0 for finite data.  An output
//     that comes out non-finite is recomputed by the faithful loop, so inf / NaN inputs still poison exactly the components
//     the reference's product poisons.
// What stays the reference's (interpolation.rs:92-131, conv_types.rs:406-424): j accumulated tap by tap in f32, the sum taken
// tap by tap in the reference's order, each weight the same expression as in k_interp_scalar_v2 (bit-identical results on
// finite data up to the sign of a zero).  *Measured*: profiles/r06_interp_frac.txt, DESIGN.md 4.4.
template <typename T>
struct FracCtx {
    const T *tabc, *tabs;
    T cb0, sb0, tworo, jsing, wsing;
    int fid;
};

template <typename T>
__device__ __forceinline__ T frac_tap_weight(const FracCtx<T>& c, T j, T sjk, int k)
{
    const T one = (T)1, two = (T)2, pi = (T)3.14159265358979323846;
    const T pi_x = pi * j;
    T w;
    if (c.fid == 0) {
        w = quot<T>(sjk, pi_x);
    } else {
        const T arg = c.tworo * j;
        const T t = one - dev_abs(arg);
        if (dev_abs(t) < (T)0.25) w = quot<T>(sjk * rc_near_num<T>(t), pi_x * (two - t));
        else {
            const T cc = c.cb0 * c.tabc[k] - c.sb0 * c.tabs[k];
            w = quot<T>(sjk * cc, pi_x * (one - arg * arg));
        }
        w = dev_abs(j) == c.jsing ? c.wsing : w;
    }
    return j == (T)0 ? one : w;
}

template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_interp_frac_pk(const T* __restrict__ x, T* __restrict__ y,
                                                         long long points_, long long new_points, int conv_len,
                                                         T factor, T delay, int fid, T rolloff,
                                                         unsigned long long slow_pairs)
{
    typedef T v2f __attribute__((ext_vector_type(2))); // (f64: the same structure on pairs of scalar operations)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ntaps = 2 * conv_len + 1;
    const int npairs = ntaps >> 1; // ntaps is odd: the last tap goes alone
    T* tabc = reinterpret_cast<T*>(smem_raw);
    T* tabs = tabc + ((ntaps + 3) & ~3);
    if (fid != 0) {
        for (int k = threadIdx.x; k < ntaps; k += 256) {
            double sk, ck;
            sincospi((double)rolloff * (double)k, &sk, &ck);
            tabc[k] = (T)ck;
            tabs[k] = (T)sk;
        }
        __syncthreads();
    }
    const int points = (int)points_;
    const T one = (T)1, two = (T)2, pi = (T)3.14159265358979323846;
    FracCtx<T> c;
    c.tabc = tabc; c.tabs = tabs; c.fid = fid;
    c.wsing = one;
    c.jsing = fid != 0 ? one / (two * rolloff) : (T)-1;
    if (fid != 0) {
        const T arg = pi / two / rolloff;
        c.wsing = dev_sin(arg) / arg * pi / (two * two);
    }
    c.tworo = two * rolloff;
    const v2f pi2 = v2f{pi, pi}, tworo2 = v2f{c.tworo, c.tworo}, one2 = v2f{one, one};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < new_points; i += (long long)gridDim.x * blockDim.x) {
        const T center = (T)i / factor;
        const T rounded = dev_floor<T>(center);
        long long p0 = ((long long)rounded - conv_len - 1) % points_;
        if (p0 < 0) p0 += points_;
        const T j0 = -(T)conv_len - (center - rounded) + delay;
        T sj, cdummy;
        dev_sincospi<T>(j0, &sj, &cdummy);
        c.cb0 = one; c.sb0 = (T)0;
        if (fid != 0) dev_sincospi<T>(rolloff * j0, &c.sb0, &c.cb0);
        // the faithful tap loop (k_interp_scalar_v2's): waves that cross the end of the vector, non-finite outputs
        auto faithful = [&](v2f& acc_out, T& sr_out) {
            int pos = (int)p0;
            T j = j0, sjk = sj;
            v2f acc = v2f{(T)0, (T)0};
            T sr = (T)0;
            for (int k = 0; k < ntaps; ++k) {
                pos = pos + 1 < points ? pos + 1 : 0;
                const T w = frac_tap_weight<T>(c, j, sjk, k);
                sjk = -sjk;
                if (CPLX) {
                    const v2f z = reinterpret_cast<const v2f*>(x)[pos];
                    const v2f zw = z * v2f{w, w}, z0 = v2f{z.y, z.x} * v2f{(T)0, (T)0};
                    acc = acc + v2f{zw.x - z0.x, zw.y + z0.y};
                } else sr = sr + x[pos] * w;
                j = j + one;
            }
            acc_out = acc;
            sr_out = sr;
        };
        v2f acc = v2f{(T)0, (T)0};
        T sr = (T)0;
        const bool wraps = p0 + ntaps >= points_;
        if (__builtin_amdgcn_ballot_w64(wraps) != 0ull) {
            faithful(acc, sr);
        } else {
            const v2f* xc = reinterpret_cast<const v2f*>(x) + (p0 + 1);
            const T* xr = x + (p0 + 1);
            const v2f sjj = v2f{sj, -sj};
            const v2f cb2 = v2f{c.cb0, c.cb0}, sb2 = v2f{c.sb0, c.sb0};
            T ja = j0;
#pragma unroll 2
            for (int p = 0; p < npairs; ++p) {
                const T jb = ja + one;
                v2f w;
                if ((slow_pairs >> p) & 1ull) { // wave-uniform: a pair that CAN hold j == 0 or a tap near / at the second singularity
                    w.x = frac_tap_weight<T>(c, ja, sj, 2 * p);
                    w.y = frac_tap_weight<T>(c, jb, -sj, 2 * p + 1);
                } else {
                    const v2f jj = v2f{ja, jb};
                    const v2f pix = pi2 * jj;
                    if (fid == 0) {
                        w = v2f{quot<T>(sjj.x, pix.x), quot<T>(sjj.y, pix.y)};
                    } else {
                        const v2f arg = tworo2 * jj;
                        const v2f tc = *reinterpret_cast<const v2f*>(tabc + 2 * p), ts = *reinterpret_cast<const v2f*>(tabs + 2 * p);
                        const v2f cc = cb2 * tc - sb2 * ts;
                        const v2f den = pix * (one2 - arg * arg);
                        const v2f num = sjj * cc;
                        w = v2f{quot<T>(num.x, den.x), quot<T>(num.y, den.y)};
                    }
                }
                if (CPLX) {
                    const v2f z0 = xc[2 * p], z1 = xc[2 * p + 1];
                    acc = acc + z0 * v2f{w.x, w.x};
                    acc = acc + z1 * v2f{w.y, w.y};
                } else {
                    sr = sr + xr[2 * p] * w.x;
                    sr = sr + xr[2 * p + 1] * w.y;
                }
                ja = jb + one;
            }
            { // the last tap (k = 2 L, even: sign +)
                const T w = frac_tap_weight<T>(c, ja, sj, ntaps - 1);
                if (CPLX) acc = acc + xc[ntaps - 1] * v2f{w, w};
                else sr = sr + xr[ntaps - 1] * w;
            }
            const T big = sizeof(T) == 4 ? (T)3.4028234e38f : (T)1.7976931348623157e308;
            const bool finite = CPLX ? (dev_abs(acc.x) <= big && dev_abs(acc.y) <= big) : dev_abs(sr) <= big;
            if (!finite) faithful(acc, sr); // inf / NaN in the data: the reference's product decides which components they reach
        }
        if (CPLX) reinterpret_cast<v2f*>(y)[i] = acc;
        else y[i] = sr;
    }
}

// the pairs of taps (2p, 2p+1) that can hold a special tap, for every output of a launch: tap k has
// j = k - L - frac + delay with frac in [0, 1).  Special: j == 0 (weight 1), and for the raised cosine |1 - |2 beta j|| < 0.25
// (the cancellation-free form and, inside it, the singularity |j| == 1 / (2 beta)).  Intervals widened by a margin far above
// the rounding of the accumulated j.  All ones = every pair takes the per-tap code (odd parameters).
static unsigned long long frac_slow_pairs(size_t conv_len, double delay, int fid, double rolloff)
{
    const long long ntaps = 2 * (long long)conv_len + 1;
    if (ntaps > 128 || !(delay == delay) || delay > 1e6 || delay < -1e6) return ~0ull;
    if (fid != 0 && !(rolloff == rolloff)) return ~0ull;
    const double beta = rolloff < 0 ? -rolloff : rolloff;
    unsigned long long mask = 0;
    for (long long k = 0; k + 1 < ntaps; ++k) {
        const double jlo = (double)k - (double)conv_len - 1.0 + delay, jhi = (double)k - (double)conv_len + delay;
        const double m = 1e-3 * (1.0 + (jhi < 0 ? -jlo : jhi));
        auto hits = [&](double a, double b) { return jlo - m < b && jhi + m > a; }; // [jlo, jhi] meets (a, b)
        bool special = hits(-0.5, 0.5);
        if (fid != 0 && beta > 0) {
            const double lo = 0.375 / beta, hi = 0.625 / beta;
            special = special || hits(lo, hi) || hits(-hi, -lo);
        }
        if (special) mask |= 1ull << (k >> 1);
    }
    return mask;
}

// scalar path with host-sampled weights (interpolatef_custom): w[i*ntaps + k] is the callback's value for tap k
// of output i, sampled on the host with the same accumulated arguments as k_interp_scalar
template <typename T, bool CPLX>
__global__ __launch_bounds__(256) void k_interp_scalar_tab(const T* __restrict__ x, T* __restrict__ y,
                                                            const T* __restrict__ w, long long points,
                                                            long long new_points, int conv_len, T factor)
{
    const int ntaps = 2 * conv_len + 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < new_points;
         i += (long long)gridDim.x * blockDim.x) {
        T center = (T)i / factor;
        T rounded = dev_floor<T>(center);
        long long pos = ((long long)rounded - conv_len - 1) % points;
        if (pos < 0) pos += points;
        const T* wi = w + i * ntaps;
        T sr = 0, si = 0;
        for (int k = 0; k < ntaps; ++k) {
            pos = pos + 1 < points ? pos + 1 : 0;
            if (CPLX) {
                T re = x[2 * pos], im = x[2 * pos + 1];
                sr = sr + (re * wi[k] - im * (T)0);
                si = si + (re * (T)0 + im * wi[k]);
            } else sr = sr + x[pos] * wi[k];
        }
        if (CPLX) { y[2 * i] = sr; y[2 * i + 1] = si; }
        else y[i] = sr;
    }
}

// The per-phase tap table of the integer-factor path, CACHED per (device, precision, function, roll-off, conv_len, factor,
// delay): a caller that interpolates vector after vector with one filter -- the normal case -- pays the table kernel
// (4.6 us of config C4b's 88) once.  Entries live in plain device memory for the life of the process (at most 32 of at
// most 60 KB); a 33rd parameter set takes the uncached route through the workspace.  The stream is synchronised once when
// an entry is created, so any stream may read it afterwards (warm the plan before a graph capture, like every other table).
struct TapKey {
    int dev, prec, fid, conv_len, factor;
    unsigned long long rolloff_bits, delay_bits;
    bool operator<(const TapKey& o) const
    {
        if (dev != o.dev) return dev < o.dev;
        if (prec != o.prec) return prec < o.prec;
        if (fid != o.fid) return fid < o.fid;
        if (conv_len != o.conv_len) return conv_len < o.conv_len;
        if (factor != o.factor) return factor < o.factor;
        if (rolloff_bits != o.rolloff_bits) return rolloff_bits < o.rolloff_bits;
        return delay_bits < o.delay_bits;
    }
};
static std::mutex g_tap_mu;
static std::map<TapKey, void*> g_tap_cache;

template <typename T>
static int interp_tap_table(int fid, T rolloff, int conv_len, int f, T delay, hipStream_t s, WsBlock* fallback, const T** table)
{
    const int ntaps = 2 * conv_len + 1;
    const size_t bytes = sizeof(T) * (size_t)ntaps * f;
    int dev = 0;
    BDSP_HIP_TRY(hipGetDevice(&dev));
    TapKey key{dev, (int)sizeof(T), fid, conv_len, f, 0, 0};
    { double r = (double)rolloff, d = (double)delay; memcpy(&key.rolloff_bits, &r, 8); memcpy(&key.delay_bits, &d, 8); }
    std::lock_guard<std::mutex> lk(g_tap_mu);
    auto it = g_tap_cache.find(key);
    if (it != g_tap_cache.end()) { *table = static_cast<const T*>(it->second); return BDSP_OK; }
    T* dst = nullptr;
    // A table is cached only when this call may synchronise the stream: a cache miss while the stream is being captured
    // into a HIP graph takes the uncached workspace route (the table kernel is then part of the graph), because
    // hipStreamSynchronize on a capturing stream invalidates the capture.
    // ... and so does a query that FAILS: the legacy null stream while another stream captures in global mode answers
    // hipErrorStreamCaptureImplicit -- exactly the case in which hipMalloc / hipStreamSynchronize would break that capture
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    bool query_ok = true;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); query_ok = false; }
    const bool cache = g_tap_cache.size() < 32 && query_ok && cap == hipStreamCaptureStatusNone;
    if (cache) BDSP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dst), bytes));
    else { BDSP_TRY(fallback->alloc(bytes, s)); dst = fallback->as<T>(); }
    hipLaunchKernelGGL((k_interp_taps<T>), dim3((f * ntaps + 63) / 64), dim3(64), 0, s, dst, fid, rolloff, conv_len, f, delay);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && cache) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        if (cache) (void)hipFree(dst); // never entered into the cache: give it back
        return hip_fail(e, "interp_tap_table", __FILE__, __LINE__);
    }
    if (cache) g_tap_cache[key] = dst;
    *table = dst;
    return BDSP_OK;
}

template <typename T>
int interpolatef_dev(const T* in, T* out, size_t len, bool is_complex, int fid, T rolloff, T factor,
                     T delay, size_t conv_len, T delta, hipStream_t s, T (*host_fn)(const void*, T),
                     const void* host_fn_data)
{
    const size_t elem = is_complex ? 2 : 1;
    const size_t points = len / elem;
    if (points == 0) return BDSP_OK;
    delay = delay / delta;                                    // interpolation.rs:397
    if (conv_len > points / 2) conv_len = points / 2;         // :399-404
    const size_t new_len = interpolatef_new_len<T>(len, factor);
    const size_t new_points = new_len / elem;
    if (new_points == 0) return BDSP_OK;
    T rf = sizeof(T) == 4 ? (T)roundf((float)factor) : (T)round((double)factor);
    T dif = rf - factor;
    if (dif < 0) dif = -dif;
    const bool simd = conv_len <= 202 && new_len >= 2000 && dif < (T)1e-6; // :411-414
    size_t blocks = (new_points + 255) / 256;
    size_t cap = (size_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (simd) {
        int f = (int)rf;
        int ntaps = 2 * (int)conv_len + 1;
        WsBlock tb;
        const T* taps_dev = nullptr;
        if (host_fn) { // the callback variant: the same table, sampled on the host
            BDSP_TRY(tb.alloc(sizeof(T) * (size_t)ntaps * f, s));
            std::vector<T> ht((size_t)ntaps * f);
            for (int sft = 0; sft < f; ++sft) {
                const T offset = (T)sft / (T)f;
                T j = -((T)conv_len - (T)1) + delay;
                for (int m = 0; m < ntaps; ++m) { ht[(size_t)sft * ntaps + m] = host_fn(host_fn_data, j - offset); j = j + (T)1; }
            }
            BDSP_HIP_TRY(hipMemcpyAsync(tb.p, ht.data(), sizeof(T) * ht.size(), hipMemcpyHostToDevice, s));
            BDSP_HIP_TRY(hipStreamSynchronize(s));
            taps_dev = tb.as<T>();
        } else {
            BDSP_TRY(interp_tap_table<T>(fid, rolloff, (int)conv_len, f, delay, s, &tb, &taps_dev));
        }
        size_t lds = sizeof(T) * (size_t)ntaps * f;
        if (lds > 60 * 1024) { set_last_error("interpolatef: tap table exceeds LDS"); return BDSP_ERR_UNSUPPORTED; }
        // edges (and everything, for factors without a blocked instantiation) by the table path; the inner region by
        // the blocked kernel where one exists -- one launch then, the edge runs taken by its last workgroups
        const long long scalar_len = (long long)ntaps * f;
        long long q_lo = (scalar_len + f - 1) / f;                   // first q with f*q >= scalar_len
        long long q_hi = ((long long)new_points - scalar_len) / f;   // f*q + f - 1 < new_points - scalar_len
        const bool blocked = (f == 2 || f == 3 || f == 4 || f == 8) && q_hi > q_lo &&
                             (long long)new_points >= 2 * scalar_len;
        if (!blocked) {
            if (is_complex)
                hipLaunchKernelGGL((k_interp_table<T, true>), dim3((unsigned)blocks), dim3(256), lds, s, in, out,
                                   taps_dev, (long long)points, (long long)new_points, (int)conv_len, f, -1LL, -1LL);
            else
                hipLaunchKernelGGL((k_interp_table<T, false>), dim3((unsigned)blocks), dim3(256), lds, s, in, out,
                                   taps_dev, (long long)points, (long long)new_points, (int)conv_len, f, -1LL, -1LL);
        } else {
            const size_t edge_outputs = new_points - (size_t)((q_hi - q_lo) * f);
            size_t eblocks = (edge_outputs + 255) / 256;
            if (eblocks > blocks) eblocks = blocks;
            if (eblocks < 1) eblocks = 1;
            const int e = is_complex ? 2 : 1;
            // positions per thread: as many as keep the output staging buffer at 32 KB
            // *measured* (round 3, tools/c4b_bench.py, 4M -> 16M): real f64 with 1 / 2 / 3 / 4 positions per thread 45.4 / 35.9 /
            // 34.9 / 37.4 us; complex f32 with 1 / 2 / 4: 40.6 / 40.0 / 45.5; complex f64 stays at 1 (round 2: 2 halves the
            // occupancy through its 32 KB staging buffer)
            constexpr int QB = sizeof(T) == 4 ? 2 : 1;   // complex
            constexpr int QBR = sizeof(T) == 4 ? 4 : 2;  // real
            const int qb = is_complex ? QB : QBR;
            const size_t tile = 256 * (size_t)qb;
            size_t lds2 = sizeof(T) * ((((size_t)f * (ntaps + 1) + 3) & ~(size_t)3) + (((tile + 2 * conv_len + 2) * e + 3) & ~(size_t)3) + tile * (size_t)f * e);
            if (lds2 < lds) lds2 = lds;
            const unsigned g = (unsigned)((q_hi - q_lo + (long long)tile - 1) / (long long)tile);
            const int stream_out = sizeof(T) * new_len > (size_t(192) << 20) ? 1 : 0; // (see the store loop of k_interp_inner)
#define BDSP_INNER2(FV, CP, QV)                                                                    \
    do {                                                                                           \
        auto kk = k_interp_inner<T, CP, FV, QV>;                                                   \
        if (lds2 > 64 * 1024)                                                                      \
            BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); \
        hipLaunchKernelGGL(kk, dim3(g + (unsigned)eblocks), dim3(256), lds2, s, in, out, taps_dev, q_lo, q_hi, (int)conv_len, \
                           (long long)points, (long long)new_points, g, stream_out);               \
    } while (0)
#define BDSP_INNER(FV)                                                                             \
    do {                                                                                           \
        if (is_complex) BDSP_INNER2(FV, true, QB);                                                 \
        else BDSP_INNER2(FV, false, QBR);                                                          \
    } while (0)
            if (f == 2) BDSP_INNER(2); else if (f == 3) BDSP_INNER(3); else if (f == 4) BDSP_INNER(4); else BDSP_INNER(8);
#undef BDSP_INNER2
#undef BDSP_INNER
        }
    } else if (host_fn) {
        const size_t ntaps = 2 * conv_len + 1;
        std::vector<T> hw(new_points * ntaps);
        for (size_t i = 0; i < new_points; ++i) {
            const T center = (T)i / factor;
            const T rounded = sizeof(T) == 4 ? (T)floorf((float)center) : (T)floor((double)center);
            T j = -(T)conv_len - (center - rounded) + delay;
            for (size_t k = 0; k < ntaps; ++k) { hw[i * ntaps + k] = host_fn(host_fn_data, j); j = j + (T)1; }
        }
        WsBlock wb;
        BDSP_TRY(wb.alloc(sizeof(T) * hw.size(), s));
        BDSP_HIP_TRY(hipMemcpyAsync(wb.p, hw.data(), sizeof(T) * hw.size(), hipMemcpyHostToDevice, s));
        BDSP_HIP_TRY(hipStreamSynchronize(s));
        if (is_complex)
            hipLaunchKernelGGL((k_interp_scalar_tab<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, wb.as<T>(),
                               (long long)points, (long long)new_points, (int)conv_len, factor);
        else
            hipLaunchKernelGGL((k_interp_scalar_tab<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, wb.as<T>(),
                               (long long)points, (long long)new_points, (int)conv_len, factor);
        BDSP_LAUNCH_CHECK();
        BDSP_HIP_TRY(hipStreamSynchronize(s)); // the weight table is released on return
        return BDSP_OK;
    } else {
        const size_t tab_bytes = (size_t)2 * (2 * conv_len + 1) * sizeof(T);
        // k_interp_scalar_v2 wherever the cos / sin table of the roll-off lattice fits 48 KB of LDS (every practical conv_len:
        // up to 3071 taps a side in f32, 1535 in f64) and positions fit 32 bits; otherwise the first-generation kernel
        // (tests/test_gpu_parity.py::test_interpolatef_fractional_factor_kernel_singularities_and_fallback runs both)
        static const bool no_pk = lab_flag("BDSP_INTERP_NO_PK"); // (LAB: A/B against k_interp_scalar_v2)
        // at most 127 taps (the slow-pair mask is one 64-bit word; above: k_interp_scalar_v2)
        if (!no_pk && 2 * conv_len + 1 <= 127 && points < ((size_t)1 << 31)) {
            const unsigned long long slow = frac_slow_pairs(conv_len, (double)delay, fid, (double)rolloff);
            const size_t lds = sizeof(T) * 2 * ((2 * conv_len + 1 + 3) & ~(size_t)3);
            if (is_complex)
                hipLaunchKernelGGL((k_interp_frac_pk<T, true>), dim3((unsigned)blocks), dim3(256), lds, s, in, out, (long long)points,
                                   (long long)new_points, (int)conv_len, factor, delay, fid, rolloff, slow);
            else
                hipLaunchKernelGGL((k_interp_frac_pk<T, false>), dim3((unsigned)blocks), dim3(256), lds, s, in, out, (long long)points,
                                   (long long)new_points, (int)conv_len, factor, delay, fid, rolloff, slow);
            BDSP_LAUNCH_CHECK();
            return BDSP_OK;
        }
        if (tab_bytes <= 48 * 1024 && points < ((size_t)1 << 31)) {
            if (is_complex)
                hipLaunchKernelGGL((k_interp_scalar_v2<T, true>), dim3((unsigned)blocks), dim3(256), tab_bytes, s, in, out,
                                   (long long)points, (long long)new_points, (int)conv_len, factor, delay, fid, rolloff);
            else
                hipLaunchKernelGGL((k_interp_scalar_v2<T, false>), dim3((unsigned)blocks), dim3(256), tab_bytes, s, in, out,
                                   (long long)points, (long long)new_points, (int)conv_len, factor, delay, fid, rolloff);
        } else if (is_complex)
            hipLaunchKernelGGL((k_interp_scalar<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, in, out,
                               (long long)points, (long long)new_points, (int)conv_len, factor, delay, fid, rolloff);
        else
            hipLaunchKernelGGL((k_interp_scalar<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, in, out,
                               (long long)points, (long long)new_points, (int)conv_len, factor, delay, fid, rolloff);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}


// ---------------------------------------------------------------------------------------------
// convolve(function, ratio, len) in the time domain: convolve_function_priv (time_freq/mod.rs:174-213)
//   y[i] = sum_{m=-L}^{L} x[(i + m) mod N] * f(-m * ratio)         (WrappingIterator pre-increments)
// The weights do not depend on i: they are tabulated once (2L+1 values, stored at `stride` scalars,
// optionally reversed so that the table is the tap vector of the centred convolution a9).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_conv_fn_taps(T* __restrict__ taps, long long L, int fid, T rolloff, T ratio, int stride,
                               bool reversed)
{
    long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k > 2 * L) return;
    T j = (T)k - (T)L; // the reference accumulates j = -L, -L+1, ...: integers, exact in T
    T w = conv_time_value<T>(fid, rolloff, -j * ratio);
    long long at = reversed ? 2 * L - k : k;
    taps[at * stride] = w;
    if (stride == 2) taps[at * 2 + 1] = (T)0;
}

template <typename T, bool CPLX, bool CW = false>
__global__ void __launch_bounds__(256)
k_conv_function(const T* __restrict__ in, T* __restrict__ out, const T* __restrict__ taps, long long points, long long L)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < points; i += (long long)gridDim.x * 256) {
        long long p = (i - L) % points;
        if (p < 0) p += points;
        T sre = 0, sim = 0;
        for (long long k = 0; k <= 2 * L; ++k) {
            T w = CW ? taps[2 * k] : taps[k];
            if (CW) { // complex weights (convolve_complex): Complex * Complex, then the sum
                const T wi = taps[2 * k + 1], xr = in[2 * p], xi = in[2 * p + 1];
                sre = sre + (xr * w - xi * wi);
                sim = sim + (xr * wi + xi * w);
            } else if (CPLX) {
                sre = sre + in[2 * p] * w;
                sim = sim + in[2 * p + 1] * w;
            } else {
                sre = sre + in[p] * w;
            }
            if (++p == points) p = 0;
        }
        if (CPLX) { out[2 * i] = sre; out[2 * i + 1] = sim; }
        else out[i] = sre;
    }
}

template <typename T>
int conv_function_taps(T* taps, size_t conv_len, int fid, T rolloff, T ratio, int stride, bool reversed, hipStream_t s)
{
    size_t n = 2 * conv_len + 1;
    hipLaunchKernelGGL((k_conv_fn_taps<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, taps,
                       (long long)conv_len, fid, rolloff, ratio, stride, reversed);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template <typename T>
int conv_function_direct(const T* in, T* out, size_t points, bool is_complex, const T* taps, size_t conv_len, hipStream_t s,
                         bool complex_taps)
{
    if (points == 0) return BDSP_OK;
    size_t blocks = (points + 255) / 256, cap = (size_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (complex_taps)
        hipLaunchKernelGGL((k_conv_function<T, true, true>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, taps,
                           (long long)points, (long long)conv_len);
    else if (is_complex)
        hipLaunchKernelGGL((k_conv_function<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, taps,
                           (long long)points, (long long)conv_len);
    else
        hipLaunchKernelGGL((k_conv_function<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, taps,
                           (long long)points, (long long)conv_len);
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

// ---------------------------------------------------------------------------------------------
// Real-only interpolation between samples (time_freq/real_interpolation.rs:33-176): linear and
// piecewise cubic Hermite (Catmull-Rom) with linearly extrapolated end points; no wrap-around.
// dest_len = round((len-1)*factor) + 1.  Evaluated in T with the reference's operation order.
// Reads the reference would panic on (a delay that pushes an index past the ends) are clamped.
// The reference accumulates the output index in T (i = i + 1), which stalls at 2^24 in f32; this
// kernel converts the integer index instead -- identical below 2^24 outputs.
// ---------------------------------------------------------------------------------------------
template <typename T>
size_t interpolate_real_len(size_t len, T factor)
{
    if (len == 0) return 0;
    T v = (T)(len - 1) * factor;
    return (size_t)(sizeof(T) == 4 ? roundf((float)v) : round((double)v)) + 1;
}

template <typename T>
__device__ __forceinline__ T clamped(const T* __restrict__ x, long long len, long long i)
{
    i = i < 0 ? 0 : (i >= len ? len - 1 : i);
    return x[i];
}

template <typename T>
__global__ void __launch_bounds__(256)
k_interp_lin(const T* __restrict__ in, T* __restrict__ out, long long len, long long dest_len, T factor, T delay)
{
    for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < dest_len; n += (long long)gridDim.x * 256) {
        if (n == dest_len - 1) { out[n] = in[len - 1]; continue; } // :68
        T rounded = (T)n / factor + delay;
        T beforef = dev_floor(rounded);
        long long before = (long long)beforef;
        T y0 = clamped(in, len, before), y1 = clamped(in, len, before + 1);
        out[n] = y0 + (y1 - y0) * (rounded - beforef);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_interp_hermite(const T* __restrict__ in, T* __restrict__ out, long long len, long long dest_len, T factor, T delay,
                 long long start, long long tail)
{
    const T half = (T)0.5, c15 = (T)1.5, two = (T)2, c25 = (T)2.5;
    for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < dest_len; n += (long long)gridDim.x * 256) {
        T rounded = (T)n / factor + delay;
        T beforef = dev_floor(rounded);
        long long before = (long long)beforef;
        T x = rounded - beforef;
        T y0, y1, y2, y3;
        if (n < start) { // :103-124
            y1 = clamped(in, len, before); y2 = clamped(in, len, before + 1); y3 = clamped(in, len, before + 2);
            y0 = y1 - (y2 - y1);
        } else if (n < tail) { // :126-145
            y0 = clamped(in, len, before - 1); y1 = clamped(in, len, before);
            y2 = clamped(in, len, before + 1); y3 = clamped(in, len, before + 2);
        } else { // :147-172
            y0 = clamped(in, len, before - 1); y1 = clamped(in, len, before);
            y2 = (before >= 0 && before < len - 1) ? in[before + 1] : y1 + (y1 - y0);
            y3 = (before >= 0 && before + 2 < len) ? in[before + 2] : y2 + (y2 - y1);
        }
        T x2 = x * x;
        T a0 = -half * y0 + c15 * y1 - c15 * y2 + half * y3;
        T a1 = y0 - c25 * y1 + two * y2 - half * y3;
        T a2 = -half * y0 + half * y2;
        T a3 = y1;
        out[n] = (a0 * x * x2) + (a1 * x2) + (a2 * x) + a3;
    }
}

template <typename T>
int interpolate_real_dev(const T* in, T* out, size_t len, T factor, T delay, bool hermite, hipStream_t s)
{
    if (len == 0) return BDSP_OK;
    const size_t dest_len = interpolate_real_len<T>(len, factor);
    size_t blocks = (dest_len + 255) / 256, cap = (size_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (!hermite) {
        hipLaunchKernelGGL((k_interp_lin<T>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, (long long)len,
                           (long long)dest_len, factor, delay);
    } else {
        T st = ((T)1 - delay) * factor;
        double c = sizeof(T) == 4 ? (double)ceilf((float)st) : ceil((double)st);
        long long start = c < 0 ? 0 : (long long)c, end = start + 1;
        if (start > (long long)dest_len) start = (long long)dest_len;
        long long tail = (long long)dest_len > end ? (long long)dest_len - end : 0;
        if (tail < start) tail = start;
        hipLaunchKernelGGL((k_interp_hermite<T>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, (long long)len,
                           (long long)dest_len, factor, delay, start, tail);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

#define BDSP_INST(T)                                                                                          \
    template int conv_function_taps<T>(T*, size_t, int, T, T, int, bool, hipStream_t);                        \
    template int conv_function_direct<T>(const T*, T*, size_t, bool, const T*, size_t, hipStream_t, bool);          \
    template size_t interpolate_real_len<T>(size_t, T);                                                       \
    template int interpolate_real_dev<T>(const T*, T*, size_t, T, T, bool, hipStream_t);
BDSP_INST(float)
BDSP_INST(double)
#undef BDSP_INST

template int interpolatef_dev<float>(const float*, float*, size_t, bool, int, float, float, float, size_t, float, hipStream_t, float (*)(const void*, float), const void*);
template int interpolatef_dev<double>(const double*, double*, size_t, bool, int, double, double, double, size_t, double, hipStream_t, double (*)(const void*, double), const void*);

} // namespace bdsp
