// mixed_radix.hip -- FFTs of lengths n = 2^a 3^b 5^c 7^d 11^e 13^f that are not powers of two (the reference takes any
// length through rustfft's mixed-radix plans, time_freq/mod.rs:47-58; its OpenCL backend accepted the 2,3,5,7,11,13
// smooth ones, ocl/mod.rs:277-299).  Bluestein (bluestein.hip) stays the path for everything else: it costs two
// power-of-two transforms of >= 2n points plus three elementwise passes, about 5x a transform of similar size.
//
//   * n <= 4096 (f32) / 2048 (f64): one workgroup-resident Stockham transform (several small transforms per workgroup),
//     ping-pong LDS buffers, radix 4/2/3/5/7/11/13 stages, stage twiddles from a copy of the cached exp(-2 pi i m / n) table in LDS;
//   * larger n = n1 * n2 (both smooth, both <= MR_PASS_MAX): four-step --
//       pass 1: tiles of W adjacent columns, FFT_n1 down the columns, x w_n^(k1 c), same layout out;
//       pass 2: W adjacent rows per workgroup, FFT_n2 along the rows, transposed store X[k1 + n1 k2]
//     -- two trips through HBM with 64-byte runs at worst (W complex values);
//   * where that tile would be a single column, or n has no such split: n = r0 r1 r2, three global Stockham passes
//     (k_mr_gpass, round 5) with tiles at most 8 wide in f32 and 4 wide in f64 (64-byte runs).
// Both take the same fused options as the power-of-two and Bluestein paths: input rotation (ifft_shift), input
// scale, window on the input or divided out of the output, real input, output rotation (fft_shift), real-part or
// magnitude output.  Unnormalised in both directions.
#include "bdsp_internal.h"
#include "dsp_funcs.h"
#include "mr_dft.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace bdsp {

constexpr int MR_MAX_STAGES = 24;
constexpr int MR_THREADS = 1024; // an LDS-bound workgroup owns its CU: sixteen waves hide the exchange latency
struct MrStages {
    int count;
    int radix[MR_MAX_STAGES];
};

// stage radices for n = 2^a 3^b 5^c 7^d 11^e 13^f: primes are paired into composite radices up to 16 so that the
// transform crosses LDS as few times as possible (1000 -> 10 10 10, 360 -> 10 12 3, 4096 -> 16 16 16)
static bool mr_factor(size_t n, MrStages* st)
{
    st->count = 0;
    int e2 = 0, e3 = 0, e5 = 0, e7 = 0, e11 = 0, e13 = 0;
    while (n % 2 == 0 && n > 1) { ++e2; n /= 2; }
    while (n % 3 == 0 && n > 1) { ++e3; n /= 3; }
    while (n % 5 == 0 && n > 1) { ++e5; n /= 5; }
    while (n % 7 == 0 && n > 1) { ++e7; n /= 7; }
    while (n % 11 == 0 && n > 1) { ++e11; n /= 11; }
    while (n % 13 == 0 && n > 1) { ++e13; n /= 13; }
    if (n != 1) return false;
    auto push = [&](int r) { if (st->count < MR_MAX_STAGES) st->radix[st->count] = r; ++st->count; };
    while (e5 > 0 && e2 > 0) { push(10); --e5; --e2; }
    while (e5 > 0 && e3 > 0) { push(15); --e5; --e3; }
    while (e7 > 0 && e2 > 0) { push(14); --e7; --e2; }
    while (e3 > 0 && e2 > 1) { push(12); --e3; e2 -= 2; }
    while (e3 > 0 && e2 > 0) { push(6); --e3; --e2; }
    while (e3 > 1) { push(9); e3 -= 2; }
    while (e2 > 3) { push(16); e2 -= 4; }
    if (e2 == 3) { push(8); e2 = 0; }
    if (e2 == 2) { push(4); e2 = 0; }
    if (e2 == 1) { push(2); e2 = 0; }
    while (e3-- > 0) push(3);
    while (e5-- > 0) push(5);
    while (e7-- > 0) push(7);
    while (e11-- > 0) push(11);
    while (e13-- > 0) push(13);
    return st->count <= MR_MAX_STAGES;
}

// One Stockham stage over `lanes` interleaved sequences of `len` points held as in[e * lanes + q]:
//   j < len/R, k = j mod ns:  v[r] = in[j + r len/R] w_{ns R}^{r k};  out[(j / ns) ns R + k + r ns] = DFT_R(v)[r]
// `tw` is the forward table exp(-2 pi i m / len).
template <int R, int DIR, typename T>
__device__ __forceinline__ void mr_stage(const cpx<T>* __restrict__ in, cpx<T>* __restrict__ out, int len, int lanes,
                                         int ns, const cpx<T>* __restrict__ tw)
{
    const int nb = len / R, tstep = len / (ns * R);
    // index arithmetic without integer divisions (two divisions and two remainders per butterfly were a quarter of a
    // radix-10 butterfly's instructions): id / lanes and j / ns by float reciprocals with one correction step each
    // (id < 2^17, j < 2^13: exact after the correction).  (A shift / mask path for power-of-two `lanes` next to the
    // general one cost the f64 resident kernel 20 bytes of reserved scratch: one branch-free form instead.)
    const float rns = 1.0f / (float)ns, rl = 1.0f / (float)lanes;
    for (int id = threadIdx.x; id < nb * lanes; id += blockDim.x) {
        int j = (int)((float)id * rl), q = id - j * lanes;
        if (q >= lanes) { q -= lanes; ++j; }
        if (q < 0) { q += lanes; --j; }
        int jq = (int)((float)j * rns), k = j - jq * ns;
        if (k >= ns) { k -= ns; ++jq; }
        if (k < 0) { k += ns; --jq; }
        cpx<T> v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = in[(j + r * nb) * lanes + q];
        if (ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = twmul<DIR>(v[r], tw[r * k * tstep]);
        }
        mr_dft<R, DIR>(v);
        const int o = jq * ns * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) out[(o + r * ns) * lanes + q] = v[r];
    }
}

// the stage twiddles are read once per butterfly input: from L2 each read cost ~1 us of latency per loop trip
// (*measured* 47 us for a 1000 x 8 tile); the workgroup keeps its own copy of the table in LDS
template <typename T>
__device__ __forceinline__ const cpx<T>* mr_table_to_lds(cpx<T>* dst, const cpx<T>* __restrict__ tw, int len)
{
    for (int i = threadIdx.x; i < len; i += blockDim.x) dst[i] = tw[i];
    return dst;
}

// all stages; returns the buffer that holds the result
template <int DIR, typename T>
__device__ __forceinline__ cpx<T>* mr_transform(cpx<T>* a, cpx<T>* b, int len, int lanes, const MrStages& st,
                                                const cpx<T>* __restrict__ tw)
{
    int ns = 1;
    for (int s = 0; s < st.count; ++s) {
        __syncthreads();
        const int R = st.radix[s];
        switch (R) {
        case 16: mr_stage<16, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 15: mr_stage<15, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 14: mr_stage<14, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 13: mr_stage<13, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 12: mr_stage<12, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 11: mr_stage<11, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 10: mr_stage<10, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 9: mr_stage<9, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 8: mr_stage<8, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 7: mr_stage<7, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 6: mr_stage<6, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 5: mr_stage<5, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 4: mr_stage<4, DIR, T>(a, b, len, lanes, ns, tw); break;
        case 3: mr_stage<3, DIR, T>(a, b, len, lanes, ns, tw); break;
        default: mr_stage<2, DIR, T>(a, b, len, lanes, ns, tw); break;
        }
        ns *= R;
        cpx<T>* t = a; a = b; b = t;
    }
    __syncthreads();
    return a;
}

// ---- fused input / output options --------------------------------------------------------------------------------
template <typename T>
struct MrIo {
    const T* in;
    T* out;
    unsigned long long n;       // points per vector
    unsigned long long rot_in;  // input element i is x[(i + rot_in) mod n]
    unsigned long long rot_out; // output element i is X[(i + rot_out) mod n]
    T in_scale;
    int in_real;                // the input holds n reals
    int out_kind;               // 0 complex, 1 real part, 2 magnitude
    int window_id;              // >= 0: multiply the input by the window ...
    int window_div;             // ... or divide the output by it
    T alpha;
};

template <typename T>
__device__ __forceinline__ cpx<T> mr_load(const MrIo<T>& io, unsigned long long vec, unsigned long long i)
{
    unsigned long long j = i + io.rot_in;
    if (j >= io.n) j -= io.n;
    T re, im;
    if (io.in_real) { re = io.in[vec * io.n + j]; im = (T)0; }
    else { const T* p = io.in + 2 * (vec * io.n + j); re = p[0]; im = p[1]; }
    T w = io.in_scale;
    if (io.window_id >= 0 && !io.window_div) w = w * window_value_sym<T>(io.window_id, io.alpha, (size_t)i, (size_t)io.n);
    return cpx<T>{re * w, im * w};
}
// stores spectrum bin k
template <typename T>
__device__ __forceinline__ void mr_store(const MrIo<T>& io, unsigned long long vec, unsigned long long k, cpx<T> z)
{
    unsigned long long i = k >= io.rot_out ? k - io.rot_out : k + io.n - io.rot_out;
    T re = z.x, im = z.y;
    if (io.window_id >= 0 && io.window_div) {
        const T w = window_value_sym<T>(io.window_id, io.alpha, (size_t)i, (size_t)io.n);
        re = re / w; im = im / w;
    }
    if (io.out_kind == 0) { T* p = io.out + 2 * (vec * io.n + i); p[0] = re; p[1] = im; }
    else if (io.out_kind == 1) io.out[vec * io.n + i] = re;
    else io.out[vec * io.n + i] = sizeof(T) == 4 ? (T)hypotf((float)re, (float)im) : (T)hypot((double)re, (double)im);
}

// ---- workgroup-resident transforms: `lanes` vectors per workgroup -----------------------------------------------
template <typename T, int DIR>
__global__ __launch_bounds__(MR_THREADS) void k_mr_wg(MrIo<T> io, MrStages st, const cpx<T>* __restrict__ tw, int lanes,
                                               unsigned long long batch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = (int)io.n;
    cpx<T>* a = reinterpret_cast<cpx<T>*>(smem_raw);
    cpx<T>* b = a + (size_t)n * lanes;
    tw = mr_table_to_lds<T>(b + (size_t)n * lanes, tw, n);
    const unsigned long long v0 = (unsigned long long)blockIdx.x * lanes;
    const int live = batch - v0 < (unsigned long long)lanes ? (int)(batch - v0) : lanes;
    for (int id = threadIdx.x; id < n * lanes; id += blockDim.x) {
        const int q = id / n, e = id % n; // consecutive threads read consecutive elements of one vector
        a[e * lanes + q] = q < live ? mr_load<T>(io, v0 + q, (unsigned long long)e) : cpx<T>{(T)0, (T)0};
    }
    cpx<T>* r = mr_transform<DIR, T>(a, b, n, lanes, st, tw);
    for (int id = threadIdx.x; id < n * lanes; id += blockDim.x) {
        const int q = id / n, e = id % n;
        if (q < live) mr_store<T>(io, v0 + q, (unsigned long long)e, r[e * lanes + q]);
    }
}

// exp(-/+ 2 pi i m / n) for the twiddles between global passes: the angle is formed in double (m / n is not a float), the
// trigonometry runs in T -- sincospif's 1e-7 is the f32 transform's own rounding, and a double sincospi per element was a
// fifth of the f32 three-pass transform of 2 * 10^7 points
template <typename T, int DIR>
__device__ __forceinline__ cpx<T> mr_unit(unsigned long long m, double two_over_n)
{
    if constexpr (sizeof(T) == 4) {
        // the angle 2 m / n lies in [0, 2): folded to (-1, 1] in double BEFORE it is rounded to float, so the rounding is at
        // most 6e-8 of half a turn instead of a whole one (1.9e-7 rad of phase per twiddle for m > n / 2 otherwise)
        double a = (double)m * two_over_n;
        if (a > 1.0) a -= 2.0;
        float sn, cs;
        sincospif((float)a, &sn, &cs);
        return cpx<T>{cs, DIR < 0 ? -sn : sn};
    } else {
        double sn, cs;
        sincospi((double)m * two_over_n, &sn, &cs);
        return cpx<T>{(T)cs, (T)(DIR < 0 ? -sn : sn)};
    }
}

// ---- four-step, pass 1: W adjacent columns c of the n1 x n2 view x[r n2 + c]; FFT_n1 over r, times w_n^(k1 c) ----
template <typename T, int DIR>
__global__ __launch_bounds__(MR_THREADS) void k_mr_pass1(MrIo<T> io, cpx<T>* __restrict__ tmp, MrStages st,
                                                  const cpx<T>* __restrict__ tw, int n1, int n2, int W)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* a = reinterpret_cast<cpx<T>*>(smem_raw);
    cpx<T>* b = a + (size_t)n1 * W;
    tw = mr_table_to_lds<T>(b + (size_t)n1 * W, tw, n1);
    const unsigned long long vec = blockIdx.y;
    const int c0 = blockIdx.x * W;
    const int live = n2 - c0 < W ? n2 - c0 : W;
    const int wsh = __ffs(W) - 1; // (W is a power of two)
    for (int id = threadIdx.x; id < n1 * W; id += blockDim.x) {
        const int q = id & (W - 1), r = id >> wsh;
        a[id] = q < live ? mr_load<T>(io, vec, (unsigned long long)r * n2 + c0 + q) : cpx<T>{(T)0, (T)0};
    }
    cpx<T>* res = mr_transform<DIR, T>(a, b, n1, W, st, tw);
    cpx<T>* tv = tmp + vec * io.n;
    const double inv = 2.0 / (double)io.n;
    for (int id = threadIdx.x; id < n1 * W; id += blockDim.x) {
        const int q = id & (W - 1), k1 = id >> wsh;
        if (q >= live) continue;
        const unsigned long long m = (unsigned long long)k1 * (unsigned long long)(c0 + q); // < n1 n2 = n
        tv[(unsigned long long)k1 * n2 + c0 + q] = cmul(res[id], mr_unit<T, DIR>(m, inv));
    }
}

// ---- pass 2: W adjacent rows k1 of tmp[k1 n2 + c]; FFT_n2 over c; X[k1 + n1 k2] ----------------------------------
template <typename T, int DIR>
__global__ __launch_bounds__(MR_THREADS) void k_mr_pass2(MrIo<T> io, const cpx<T>* __restrict__ tmp, MrStages st,
                                                  const cpx<T>* __restrict__ tw, int n1, int n2, int W)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* a = reinterpret_cast<cpx<T>*>(smem_raw);
    cpx<T>* b = a + (size_t)n2 * W;
    tw = mr_table_to_lds<T>(b + (size_t)n2 * W, tw, n2);
    const unsigned long long vec = blockIdx.y;
    const int r0 = blockIdx.x * W;
    const int live = n1 - r0 < W ? n1 - r0 : W;
    const cpx<T>* tv = tmp + vec * io.n;
    for (int q = 0; q < W; ++q) // unit stride along a row
        for (int c = threadIdx.x; c < n2; c += blockDim.x)
            a[c * W + q] = q < live ? tv[(unsigned long long)(r0 + q) * n2 + c] : cpx<T>{(T)0, (T)0};
    cpx<T>* res = mr_transform<DIR, T>(a, b, n2, W, st, tw);
    const int wsh = __ffs(W) - 1; // (W is a power of two)
    for (int id = threadIdx.x; id < n2 * W; id += blockDim.x) {
        const int q = id & (W - 1), k2 = id >> wsh; // W adjacent k1 are adjacent bins
        if (q < live) mr_store<T>(io, vec, (unsigned long long)(r0 + q) + (unsigned long long)n1 * k2, res[id]);
    }
}

// ---- three global Stockham passes (round 5): n = R1 R2 R3, each pass an mr_stage whose butterfly is a whole RP-point
// mixed-radix transform held in LDS -- the structure of the power-of-two k_fft_pass with any smooth super-radix:
//     column j < n / RP, k = j mod nsg:  v[r] = in[j + r n / RP] w_n^(r k n / (nsg RP));  out[(j / nsg) nsg RP + k + r nsg] = DFT_RP(v)[r]
// A tile is W adjacent columns: loads are runs of W values, stores runs of RP values in the first pass (nsg = 1: whole
// output columns) and of W values afterwards.  With three factors of a few hundred points the tile stays at mr_tile<T>() = 8
// columns in f32 and 4 in f64 (64-byte runs; it only ever narrows from there) where the four-step form of the same length
// is down to 2- or 1-wide tiles.
// POS: 0 first pass (fused input options: mr_load), 1 middle, 2 last (fused output options: mr_store) -- compile-time, so
// that no instantiation carries both option paths (with both, hipcc reserved 52-68 bytes of scratch per lane for scalar
// spills it then kept in vector lanes after all: no kernel of the library has a private segment).
template <typename T, int DIR, int POS>
__global__ __launch_bounds__(MR_THREADS) void k_mr_gpass(MrIo<T> io, const cpx<T>* __restrict__ src, cpx<T>* __restrict__ dst,
                                                  MrStages st, const cpx<T>* __restrict__ tw, unsigned long long n, int RP,
                                                  unsigned long long nsg, int W)
{
    constexpr bool first = POS == 0, last = POS == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx<T>* a = reinterpret_cast<cpx<T>*>(smem_raw);
    cpx<T>* b = a + (size_t)RP * W;
    tw = mr_table_to_lds<T>(b + (size_t)RP * W, tw, RP);
    const unsigned long long vec = blockIdx.y, cols = n / RP, j0 = (unsigned long long)blockIdx.x * W;
    const int live = cols - j0 < (unsigned long long)W ? (int)(cols - j0) : W;
    const cpx<T>* sv = src + vec * n;
    const unsigned long long tstep = n / (nsg * RP);
    const double inv = 2.0 / (double)n;
    // (no per-element divisions: W is a power of two, and a tile's W columns j0 + q cross a multiple of nsg at most once
    // -- nsg is 1 or a whole first factor, never less than W -- so j / nsg and j mod nsg follow from the tile's own pair)
    const int wsh = __ffs(W) - 1;
    const unsigned long long jd0 = j0 / nsg, k0 = j0 % nsg;
    for (int id = threadIdx.x; id < RP * W; id += blockDim.x) {
        const int q = id & (W - 1), r = id >> wsh;
        cpx<T> v{(T)0, (T)0};
        if (q < live) {
            const unsigned long long j = j0 + q, i = j + (unsigned long long)r * cols;
            v = first ? mr_load<T>(io, vec, i) : sv[i];
            if (nsg > 1 && r > 0) {
                const unsigned long long kk = k0 + q >= nsg ? k0 + q - nsg : k0 + q;
                const unsigned long long e = (unsigned long long)r * kk * tstep; // < n: r < RP, k < nsg
                if (e) v = cmul(v, mr_unit<T, DIR>(e, inv));
            }
        }
        a[id] = v;
    }
    cpx<T>* res = mr_transform<DIR, T>(a, b, RP, W, st, tw);
    cpx<T>* dv = dst + vec * n;
    if (nsg == 1) { // whole output columns: r fastest
        for (int id = threadIdx.x; id < RP * W; id += blockDim.x) {
            const int r = id % RP, q = id / RP;
            if (q >= live) continue;
            const unsigned long long o = (j0 + q) * RP + r;
            if (last) mr_store<T>(io, vec, o, res[r * W + q]); else dv[o] = res[r * W + q];
        }
    } else {
        for (int id = threadIdx.x; id < RP * W; id += blockDim.x) {
            const int q = id & (W - 1), r = id >> wsh;
            if (q >= live) continue;
            const bool wrap = k0 + q >= nsg;
            const unsigned long long kk = wrap ? k0 + q - nsg : k0 + q, jd = wrap ? jd0 + 1 : jd0;
            const unsigned long long o = jd * nsg * RP + kk + (unsigned long long)r * nsg;
            if (last) mr_store<T>(io, vec, o, res[id]); else dv[o] = res[id];
        }
    }
}

// ---- planning and launch ------------------------------------------------------------------------------------------
constexpr size_t MR_LDS_BYTES = 144 * 1024; // of the CU's 160 KB; one workgroup per CU either way

template <typename T> static size_t mr_wg_max() { return sizeof(T) == 4 ? 4096 : 2048; } // 2 buffers + the table: 96 KB
template <typename T> static int mr_tile() { return sizeof(T) == 4 ? 8 : 4; }                    // 64-byte runs
// longest sub-transform a tile of W lanes can hold: 2 ping-pong buffers of len * W points + the table
template <typename T> static size_t mr_pass_max(int W) { return MR_LDS_BYTES / (sizeof(cpx<T>) * (2 * W + 1)); }

// Four-step or three Stockham passes?  *Measured* (round 5, tools/plan_probe.py --points, cold us, LAB switch BDSP_MR_PLAN;
// profiles/r05_plan_probe_valid.txt run 13, after the index arithmetic lost its divisions): while the four-step's tile is 8
// wide it wins (10^6 points f32 32 against 49 us, f64 34 against 65 at its widest tile of 4; 64 x 10^5 f32 126 against 184);
// at 4-wide f32 tiles it depends on the factors (3 * 10^6 79 against 87, 1.5 * 10^6 60 against 54) and the four-step keeps
// them; at 2-wide tiles three passes win or tie everywhere -- f32 6 * 10^6 210 -> 180, 10^7 356 -> 280, 1.296 * 10^7 426 -> 307;
// f64 1.2 * 10^6 64.7 -> 64.1, 1.5 * 10^6 78.8 -> 75.2, 2 * 10^6 103 -> 84, 2.4 * 10^6 117 -> 105, 3.24 * 10^6 147 -> 140 -- and where the
// four-step would need single columns or has no split at all they are the only global form: f64 3 * 10^6 130 us (single
// columns 236, chirp-z 400), 6 * 10^6 255 (450, 861), 10^7 424 (chirp-z 1822); f32 2 * 10^7 541 (1115, 2019), 3 * 10^7 825 (1613, 2083).
template <typename T> static int mr_plan2_min_tile() { return 4; }
constexpr size_t MR_PLAN3_MIN = 100000;

// n1 * n2 = n with both factors smooth: the most balanced split, at the widest tile (W_max, W_max/2, ... 2) whose
// LDS holds both factors (n <= 1024^2 at the full 64-byte tile, up to ~13M points with 2-wide tiles)
template <typename T>
static bool mr_split(size_t n, size_t* n1, size_t* n2, int* wmax)
{
    // (LAB, BDSP_MR_W1: also single-column tiles -- measured in round 5: they beat the chirp-z path that served such lengths
    // until then (f64 3 * 10^6 points 400 -> 236 us) and lose to three Stockham passes (130 us), which is what runs now)
    static const bool w1 = lab_flag("BDSP_MR_W1");
    for (int W = mr_tile<T>(); W >= (w1 ? 1 : 2); W /= 2) {
        const size_t pm = mr_pass_max<T>(W);
        if (n > pm * pm) continue;
        size_t best = 0;
        for (size_t d = 2; d * d <= n; ++d) {
            if (n % d) continue;
            MrStages s;
            if (n / d <= pm && mr_factor(d, &s) && mr_factor(n / d, &s)) best = d;
        }
        if (best) { *n1 = best; *n2 = n / best; *wmax = W; return true; }
    }
    return false;
}

// n = r0 r1 r2, all smooth, for the three-pass Stockham form: the most balanced split (smallest largest factor), at the
// widest tile W in {8, 4, 2} (f32; {4, 2} in f64: it starts at mr_tile<T>() and only narrows) whose LDS holds that factor; the largest factor goes first.
template <typename T>
static bool mr_split3(size_t n, size_t r[3], int* wmax)
{
    std::vector<size_t> divs;
    for (size_t d = 1; d * d <= n; ++d)
        if (n % d == 0) { divs.push_back(d); if (d != n / d) divs.push_back(n / d); }
    std::sort(divs.begin(), divs.end());
    size_t best = 0, b0 = 0, b1 = 0, b2 = 0;
    MrStages st;
    const size_t cap = mr_pass_max<T>(4);
    for (size_t d0 : divs) {
        if (d0 < 2 || d0 > cap) continue;
        if (!mr_factor(d0, &st)) continue;
        const size_t m = n / d0;
        for (size_t d1 : divs) {
            if (d1 < 2 || d1 > d0 || m % d1) continue; // d0 >= d1 >= d2
            const size_t d2 = m / d1;
            if (d2 < 2 || d2 > d1) continue;
            if (!mr_factor(d1, &st) || !mr_factor(d2, &st)) continue;
            if (best == 0 || d0 < best) { best = d0; b0 = d0; b1 = d1; b2 = d2; }
        }
    }
    if (!best) return false;
    r[0] = b0; r[1] = b1; r[2] = b2;
    // (*measured*, BDSP_MR_W3: f64 3 / 6 / 10 * 10^6 points 157 / 315 / 499 us at 16-wide tiles, 172 / 318 / 473 at 8, 143 / 282 / 460 at
    // 4; f32 2 / 3 * 10^7 714 / 1003 at 16, 627 / 936 at 8, 657 / 992 at 4: the 64-byte tile of the four-step form)
    int W = mr_tile<T>();
    while (W > 2 && mr_pass_max<T>(W) < b0) W /= 2;
    *wmax = W;
    return true;
}

// which global form serves n (above the workgroup-resident range): 2 = four-step, 3 = three Stockham passes, 0 = neither.
// Three passes take over where the four-step's tile would be narrower than 64 bytes in f32 / ... (measured: see mr_fft).
struct MrPlan { int plan; size_t n1, n2, r3[3]; int w2, w3; };
template <typename T>
static int mr_global_plan(size_t n, size_t* n1, size_t* n2, int* w2, size_t r3[3], int* w3)
{
    // (the splits enumerate divisors: remembered per length -- a handle asks three times per transform)
    static std::mutex mu;
    static std::unordered_map<size_t, MrPlan> cache;
    MrPlan p{};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(n);
        if (it != cache.end()) p = it->second;
        else {
            static const int forced = [] { const char* e = lab_env("BDSP_MR_PLAN"); return e ? atoi(e) : 0; }();
            const bool ok2 = mr_split<T>(n, &p.n1, &p.n2, &p.w2);
            const bool ok3 = n >= (size_t)MR_PLAN3_MIN && mr_split3<T>(n, p.r3, &p.w3);
            if (forced == 2 && ok2) p.plan = 2;
            else if (forced == 3 && ok3) p.plan = 3;
            else if (ok2 && (p.w2 >= mr_plan2_min_tile<T>() || !ok3)) p.plan = 2;
            else if (ok3) p.plan = 3;
            else p.plan = ok2 ? 2 : 0;
            if (cache.size() < 4096) cache[n] = p;
        }
    }
    *n1 = p.n1; *n2 = p.n2; *w2 = p.w2; *w3 = p.w3;
    r3[0] = p.r3[0]; r3[1] = p.r3[1]; r3[2] = p.r3[2];
    return p.plan;
}

template <typename T>
bool mr_supported(size_t n)
{
    MrStages s;
    if (n < 2 || is_pow2(n) || !mr_factor(n, &s)) return false;
    if (n <= mr_wg_max<T>()) return twiddle_table_available<T>((int)n);
    size_t n1, n2, r3[3];
    int w2, w3;
    const int plan = mr_global_plan<T>(n, &n1, &n2, &w2, r3, &w3);
    if (plan == 2) return twiddle_table_available<T>((int)n1) && twiddle_table_available<T>((int)n2);
    if (plan == 3) return twiddle_table_available<T>((int)r3[0]) && twiddle_table_available<T>((int)r3[1]) && twiddle_table_available<T>((int)r3[2]);
    return false;
}

// trips through global memory: 1 resident, 2 four-step (in -> scratch -> out), 3 Stockham passes (in -> scratch -> in -> out)
template <typename T>
int mr_passes(size_t n)
{
    if (n <= mr_wg_max<T>()) return 1;
    size_t n1, n2, r3[3];
    int w2, w3;
    return mr_global_plan<T>(n, &n1, &n2, &w2, r3, &w3);
}
template int mr_passes<float>(size_t);
template int mr_passes<double>(size_t);

template <typename T> bool mr_resident(size_t n) { return n <= mr_wg_max<T>(); }
template bool mr_resident<float>(size_t);
template bool mr_resident<double>(size_t);

// threads per workgroup: about eight points of the tile per thread (a radix-8..16 butterfly each), 64 ... 1024
static unsigned mr_threads(size_t tile_points)
{
    size_t t = ((tile_points / 8 + 63) / 64) * 64;
    return (unsigned)(t < 64 ? 64 : (t > (size_t)MR_THREADS ? (size_t)MR_THREADS : t));
}

template <typename K>
static int mr_set_lds(K kern, size_t lds)
{
    if (lds > 64 * 1024)
        BDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return BDSP_OK;
}

// k_mr_reg3 (mixed_radix_reg3.h, instantiated in mixed_radix_reg3_f32.hip / _f64.hip): returns MR_REG3_NOT_BUILT when the
// length has no instantiation (the caller then takes k_mr_wg)
constexpr int MR_REG3_NOT_BUILT = 1 << 20;
template <typename T>
struct MrReg3Io { // (layout: mixed_radix_reg3.h)
    const T* in;
    T* out;
    unsigned rot_in, rot_out;
    T in_scale;
    int in_real, out_kind, plain, window_id, window_div;
    T alpha;
};
template <typename T>
int mr_reg3_launch(const MrReg3Io<T>& io, size_t n, size_t batch, bool inverse, hipStream_t s);
template <typename T>
int mr_reg2_launch(const MrReg3Io<T>& io, size_t n, size_t batch, bool inverse, hipStream_t s); // (mixed_radix_reg2.h: n < 300)

// in -> out (may alias for the workgroup-resident path; the four-step path needs `scratch` of n * batch complex)
template <typename T>
int mr_fft(const T* in, T* out, T* scratch, size_t n, size_t batch, bool inverse, unsigned flags, T in_scale,
           int window_id, T window_alpha, hipStream_t s)
{
    MrIo<T> io{};
    io.in = in; io.out = out; io.n = n;
    io.rot_in = (flags & BDSP_FFT_SHIFT_IN) ? n / 2 : 0;
    io.rot_out = (flags & BDSP_FFT_SHIFT_OUT) ? n - n / 2 : 0;
    io.in_scale = in_scale;
    io.in_real = (flags & FFT_IN_REAL) ? 1 : 0;
    io.out_kind = (flags & BDSP_FFT_MAGNITUDE) ? 2 : ((flags & FFT_OUT_REAL) ? 1 : 0);
    io.window_id = window_id;
    io.window_div = (flags & FFT_WINDOW_OUT_DIV) ? 1 : 0;
    io.alpha = window_alpha;
    if (batch == 0) return BDSP_OK;
    if (n <= mr_wg_max<T>()) {
        // the lengths k_mr_reg3 is built for, any batch, every fused option: register-resident, persistent
        {
            MrReg3Io<T> r{};
            r.in = in; r.out = out;
            r.rot_in = (unsigned)io.rot_in; r.rot_out = (unsigned)io.rot_out;
            r.in_scale = in_scale; r.in_real = io.in_real; r.out_kind = io.out_kind;
            r.window_id = window_id; r.window_div = io.window_div; r.alpha = window_alpha;
            r.plain = (io.rot_in == 0 && io.rot_out == 0 && in_scale == (T)1 && !io.in_real && io.out_kind == 0 && window_id < 0) ? 1 : 0;
            int c = MR_REG3_NOT_BUILT;
            if (n >= 300) c = mr_reg3_launch<T>(r, n, batch, inverse, s);
            else if constexpr (sizeof(T) == 4) c = mr_reg2_launch<T>(r, n, batch, inverse, s); // (f64: measured slower than k_mr_wg)
            if (c != MR_REG3_NOT_BUILT) return c;
        }
        MrStages st;
        if (!mr_factor(n, &st)) return BDSP_ERR_UNSUPPORTED;
        const cpx<T>* tw;
        BDSP_TRY(twiddle_table<T>((int)n, &tw));
        size_t lanes = 2048 / n; // small transforms share a workgroup
        if (lanes < 1) lanes = 1;
        if (lanes > 16) lanes = 16;
        if (lanes > batch) lanes = batch;
        const size_t lds = sizeof(cpx<T>) * (2 * n * lanes + n);
        const unsigned grid = (unsigned)((batch + lanes - 1) / lanes);
        // many small transforms: 256-thread workgroups, several per CU, interleave their barriers; a few large ones:
        // 1024 threads each
        const unsigned threads = grid >= 2u * (unsigned)num_cus() ? (mr_threads(n * lanes) > 256u ? 256u : mr_threads(n * lanes)) : mr_threads(4 * n * lanes);
        if (inverse) {
            BDSP_TRY(mr_set_lds(k_mr_wg<T, 1>, lds));
            hipLaunchKernelGGL((k_mr_wg<T, 1>), dim3(grid), dim3(threads), lds, s, io, st, tw, (int)lanes, (unsigned long long)batch);
        } else {
            BDSP_TRY(mr_set_lds(k_mr_wg<T, -1>, lds));
            hipLaunchKernelGGL((k_mr_wg<T, -1>), dim3(grid), dim3(threads), lds, s, io, st, tw, (int)lanes, (unsigned long long)batch);
        }
        BDSP_LAUNCH_CHECK();
        return BDSP_OK;
    }
    size_t n1, n2, r3[3];
    int wmax = 0, w3 = 0;
    const int plan = mr_global_plan<T>(n, &n1, &n2, &wmax, r3, &w3);
    if (plan == 0 || batch > 65535) return BDSP_ERR_UNSUPPORTED;
    if (plan == 3) {
        // in -> scratch -> (the input's buffer, dead by then) -> out; the caller passes out = scratch (mr_passes == 3)
        cpx<T>* bufs[2] = {reinterpret_cast<cpx<T>*>(scratch), reinterpret_cast<cpx<T>*>(const_cast<T*>(in))};
        if (out != scratch) { set_last_error("mixed radix, three passes: the result goes to the scratch buffer"); return BDSP_ERR_UNSUPPORTED; }
        int W = w3;
        static const int w3_env = [] { const char* e = lab_env("BDSP_MR_W3"); return e ? atoi(e) : 0; }();
        if (w3_env > 0 && w3_env <= w3) W = w3_env;
        while (W & (W - 1)) W &= W - 1;
        unsigned long long nsg = 1;
        const cpx<T>* src = nullptr;
        for (int p = 0; p < 3; ++p) {
            const size_t RP = r3[p];
            MrStages sp;
            mr_factor(RP, &sp);
            const cpx<T>* twp;
            BDSP_TRY(twiddle_table<T>((int)RP, &twp));
            const size_t lds = sizeof(cpx<T>) * (2 * RP * W + RP);
            cpx<T>* dst = p == 2 ? nullptr : bufs[p];
            const dim3 g((unsigned)((n / RP + W - 1) / W), (unsigned)batch);
            const unsigned th = mr_threads(2 * RP * W);
#define BDSP_GPASS(DIRV, POSV)                                                                                          \
    do {                                                                                                                \
        BDSP_TRY(mr_set_lds(k_mr_gpass<T, DIRV, POSV>, lds));                                                           \
        hipLaunchKernelGGL((k_mr_gpass<T, DIRV, POSV>), g, dim3(th), lds, s, io, src, dst, sp, twp, (unsigned long long)n, (int)RP, nsg, W); \
    } while (0)
            if (inverse) { if (p == 0) BDSP_GPASS(1, 0); else if (p == 1) BDSP_GPASS(1, 1); else BDSP_GPASS(1, 2); }
            else { if (p == 0) BDSP_GPASS(-1, 0); else if (p == 1) BDSP_GPASS(-1, 1); else BDSP_GPASS(-1, 2); }
#undef BDSP_GPASS
            BDSP_LAUNCH_CHECK();
            src = dst;
            nsg *= RP;
        }
        return BDSP_OK;
    }
    MrStages s1, s2;
    mr_factor(n1, &s1);
    mr_factor(n2, &s2);
    const cpx<T>*tw1, *tw2;
    BDSP_TRY(twiddle_table<T>((int)n1, &tw1));
    BDSP_TRY(twiddle_table<T>((int)n2, &tw2));
    // tile width: W adjacent columns / rows per workgroup (64-byte runs); a lone transform narrows the tile until
    // every CU has a workgroup -- each tile is a chain of dependent LDS stages, so latency, not bandwidth, rules
    int W = wmax;
    const int wmin = wmax < 2 ? 1 : (n >= 200000 ? 4 : 2); // *measured* 10^6 points: W = 8 / 4 / 2 -> 45.7 / 36.5 / 48.5 us; 10^5: 18.0 / 14.7 / 14.1
    while (W > wmin && ((n2 + W - 1) / W) * batch < (size_t)num_cus()) W /= 2;
    static const int w_env = [] { const char* e = lab_env("BDSP_MR_W"); return e ? atoi(e) : 0; }();
    if (w_env > 0 && w_env <= wmax) W = w_env;
    while (W & (W - 1)) W &= W - 1; // (the kernels index a tile by shifts: a power of two)
    const size_t lds1 = sizeof(cpx<T>) * (2 * n1 * W + n1), lds2 = sizeof(cpx<T>) * (2 * n2 * W + n2);
    cpx<T>* tmp = reinterpret_cast<cpx<T>*>(scratch);
    // a lone transform wants every wave it can get per tile; a batch has parallelism across tiles already
    const bool lone = ((n2 + W - 1) / W) * batch < 4 * (size_t)num_cus();
    const unsigned t1 = lone ? mr_threads(4 * n1 * W) : mr_threads(n1 * W), t2 = lone ? mr_threads(4 * n2 * W) : mr_threads(n2 * W);
    const dim3 g1((unsigned)((n2 + W - 1) / W), (unsigned)batch), g2((unsigned)((n1 + W - 1) / W), (unsigned)batch);
    if (inverse) {
        BDSP_TRY(mr_set_lds(k_mr_pass1<T, 1>, lds1));
        BDSP_TRY(mr_set_lds(k_mr_pass2<T, 1>, lds2));
        hipLaunchKernelGGL((k_mr_pass1<T, 1>), g1, dim3(t1), lds1, s, io, tmp, s1, tw1, (int)n1, (int)n2, W);
        hipLaunchKernelGGL((k_mr_pass2<T, 1>), g2, dim3(t2), lds2, s, io, tmp, s2, tw2, (int)n1, (int)n2, W);
    } else {
        BDSP_TRY(mr_set_lds(k_mr_pass1<T, -1>, lds1));
        BDSP_TRY(mr_set_lds(k_mr_pass2<T, -1>, lds2));
        hipLaunchKernelGGL((k_mr_pass1<T, -1>), g1, dim3(t1), lds1, s, io, tmp, s1, tw1, (int)n1, (int)n2, W);
        hipLaunchKernelGGL((k_mr_pass2<T, -1>), g2, dim3(t2), lds2, s, io, tmp, s2, tw2, (int)n1, (int)n2, W);
    }
    BDSP_LAUNCH_CHECK();
    return BDSP_OK;
}

template bool mr_supported<float>(size_t);
template bool mr_supported<double>(size_t);
template int mr_fft<float>(const float*, float*, float*, size_t, size_t, bool, unsigned, float, int, float, hipStream_t);
template int mr_fft<double>(const double*, double*, double*, size_t, size_t, bool, unsigned, double, int, double, hipStream_t);

} // namespace bdsp
