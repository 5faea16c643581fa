// Kernel lab: a TWO-trip 2^24-point complex f32 FFT whose extra exchange stays in the XCD's L2.
//
// The library runs 2^24 = 256 x 256 x 256 as three global Stockham passes (fft_impl.h: k_fft_pass), i.e. six
// crossings of the XCD <-> memory fabric at its ~6.5 TB/s ceiling.  This plan makes FOUR crossings:
//     n = r*4096 + c,  r = a*16 + b,  c = d*256 + e          k = kr + 4096*kc,  kr = ka + 256*kb,  kc = kd + 16*ke
//   trip 1 (one persistent launch), column groups of 16 adjacent c (128-byte runs):
//     P1 (group, b):  256-point transform over a of 16 columns (the library's first-pass tile: rows 512 KB apart)
//                     -> ring[b][c][ka]   (a 512 KB slot of a per-XCD ring that lives in that XCD's L2)
//     P2 (group, 32 ka): register 16-point transform over b with the inner twiddle W_4096^(b ka) riding on the
//                     multiply-adds (dft16_tw), times W_N^(c kr)  -> mid'[kr/16][c][kr%16]   (2 KB runs)
//   trip 2, row groups of 16 adjacent kr (512 KB contiguous in mid'):
//     P1 (group, 32 e): register 16-point transform over d, times W_4096^(e kd)  -> ring[kd][e][kr%16]
//     P2 (group, kd):  256-point transform over e of 16 columns (kr%16) -> X[kr + 4096 (kd + 16 ke)]
//                     (the library's last-pass store: 128-byte runs 512 KB apart)
// Scheduling: workgroups read HW_REG_XCC_ID and pull items from THEIR XCD's queue (one returning atomic per
// item); an XCD's item sequence interleaves the P1 tiles of its q-th group with the P2 tiles of its (q-L)-th, groups
// are claimed from one global counter (no assumption about block -> XCD placement or about balance).  An item only
// ever waits for items drawn EARLIER from the same queue, which running workgroups own: no co-residency requirement,
// plain launch.  Ring hand-off inside an XCD: plain stores + s_waitcnt vmcnt(0) + relaxed agent counter; readers use
// sc1 loads (L1 bypass, L2-served).  Every spin is bounded.
//
// Checker: all 2^24 bins against an f64 radix-2 transform on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fft_core.h"
#include "basic_dsp_hip.h"

using namespace bdsp;
typedef bdsp_f32x2 f2;
typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef LAB_WAVES
#define LAB_WAVES 4
#endif

constexpr int CS = 273;                   // col_stride(256, 16)
constexpr unsigned END = 0x7fffffffu;
constexpr unsigned SPIN_LIMIT = 1u << 24;
constexpr int MAXR = 8;
constexpr size_t SLOT = 65536;            // complex points per ring slot (512 KB)
constexpr size_t NPTS = size_t(1) << 24;

struct WgRec { unsigned long long t0, t1, wf, draw, wg, p1, p2; unsigned xcc, np1, np2, pad; };
struct Ctl {
    unsigned head[8][32];          // per-XCD item counter, one 128-byte line each
    unsigned cnt1[8][MAXR][32];    // P1 tiles finished, per ring slot (monotonic over the slot's reuse)
    unsigned cnt2[8][MAXR][32];    // P2 tiles that have finished READING the slot
    unsigned grp[8][320];          // global group id + 1 of the XCD's q-th group (0 = not claimed yet)
    unsigned next_group[32];
    unsigned exit_count[32];
    unsigned error;
    unsigned maxspin;
    unsigned pad[30];
    unsigned stats[8];             // groups per XCD of the last launch (never reset by the kernel)
    unsigned long long span[4];    // LAB_STATS: min start, max end, max alive, (unused)
    unsigned long long tm[8];      // LAB_STATS: 10 ns ticks summed over workgroups: draw, wait-group, wait-flag, P1 body, P2 body, items P1, items P2, total
};

__device__ __forceinline__ unsigned ld_flag(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, Ctl* ctl)
{
    unsigned spins = 0;
#ifdef LAB_NOWAIT
    return true;
#endif
    while (ld_flag(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { atomicOr(&ctl->error, 1u); return false; }
    }
#ifdef LAB_STATS
    if (spins) atomicMax(&ctl->maxspin, spins);
#endif
    return true;
}
__device__ __forceinline__ unsigned wait_nonzero(unsigned* p, Ctl* ctl)
{
    unsigned spins = 0, v;
    while ((v = ld_flag(p)) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { atomicOr(&ctl->error, 2u); return END + 1; }
    }
    return v;
}

// L1-bypassing (sc1) loads of ring data another CU of this XCD wrote
__device__ __forceinline__ f2 ld_sc1(const f2* p)
{
#ifdef LAB_PLAINLD
    return *p;
#endif
    unsigned long long raw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(f2, raw);
}
#ifdef LAB_BUFLOAD
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f4 ld_sc1_16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off)
{
    i4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16 /* sc1 */);
    return __builtin_bit_cast(f4, r);
}
#endif
__device__ __forceinline__ f4 ld_sc1_f4(const f2* p)
{
    f2 a = ld_sc1(p), b = ld_sc1(p + 1);
    return f4{a.x, a.y, b.x, b.y};
}

__device__ __forceinline__ f2 root24(unsigned e) // exp(-2 pi i e / 2^24), e < 2^24 exact in float
{
    float s, c;
    sincospif((float)e * (2.0f / 16777216.0f), &s, &c);
    return f2{c, -s};
}
// p[k] = s^k, k = 0..15, products at most four deep
__device__ __forceinline__ void powers16(f2 s, f2* p)
{
    p[0] = f2{1.0f, 0.0f};
    p[1] = s;
    p[2] = cmul(s, s);
    p[4] = cmul(p[2], p[2]);
    p[8] = cmul(p[4], p[4]);
    p[3] = cmul(p[2], s);
    p[5] = cmul(p[4], s);
    p[6] = cmul(p[4], p[2]);
    p[7] = cmul(p[4], p[3]);
#pragma unroll
    for (int k = 1; k < 8; ++k) p[8 + k] = cmul(p[8], p[k]);
}

template <int TRIP, int DIR>
__global__ __launch_bounds__(256, LAB_WAVES) void k_trip(const f2* __restrict__ in, f2* __restrict__ out, f2* ring, Ctl* ctl,
                                                          const f2* __restrict__ wtab, int L, int R, WgRec* rec = nullptr)
{
    using F = WgFft<float, 256, 16>;
    constexpr int NP1 = TRIP == 1 ? 16 : 8, NP2 = 24 - NP1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f2* lds = reinterpret_cast<f2*>(smem_raw);
    f2* ltw = lds + 16 * CS;
    __shared__ unsigned s_kind, s_gid, s_tile, s_slot;
    const int tid = threadIdx.x;
    ltw[tid] = wtab[tid];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    f2* const myring = ring + (size_t)xcc * R * SLOT;
    auto tw = [&](int m) { return ltw[m]; };
    __syncthreads();
#ifdef LAB_NORING
    const bool ring_on = L < 0; // ablation: no ring traffic at all (the compiler cannot know)
#else
    const bool ring_on = true;
#endif

#ifdef LAB_STATS
    unsigned long long tm_draw = 0, tm_wg = 0, tm_wf = 0, tm_p1 = 0, tm_p2 = 0, n_p1 = 0, n_p2 = 0, t_start = wall_clock64(), t_a = 0, t_b = 0;
#define TICK(x) x = wall_clock64()
#else
#define TICK(x)
#endif
    for (;;) {
        __syncthreads(); // the previous item's readers of s_* and of the LDS tile are done
        if (tid == 0) {
            TICK(t_a);
            const unsigned i = __hip_atomic_fetch_add(&ctl->head[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned q = i / 24, s = i % 24;
            if (s == 0) { // this drawer claims the XCD's NEXT group (and, at the very start, its first)
                for (unsigned qq = (q == 0 ? 0 : q + 1); qq <= q + 1; ++qq) {
                    unsigned g = __hip_atomic_fetch_add(&ctl->next_group[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (g < 256) atomicAdd(&ctl->stats[xcc], 1u);
                    __hip_atomic_store(&ctl->grp[xcc][qq], g < 256 ? g + 1 : END + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#ifdef LAB_STATS
            if (i + 1 == 0) __builtin_trap(); // (uses the returned value: the draw has completed)
            TICK(t_b); tm_draw += t_b - t_a;
#endif
            bool p2;
            unsigned tile;
            if (TRIP == 1) { p2 = (s % 3) == 2; tile = p2 ? s / 3 : (s / 3) * 2 + (s % 3); }
            else { p2 = (s % 3) != 0; tile = p2 ? (s / 3) * 2 + (s % 3) - 1 : s / 3; }
            unsigned kind = 0, gid = END, slot = 0;
            const int lq = p2 ? (int)q - L : (int)q;
            bool ok = true;
            if (lq >= 0) {
                gid = wait_nonzero(&ctl->grp[xcc][lq], ctl) - 1;
#ifdef LAB_STATS
                TICK(t_a); tm_wg += t_a - t_b;
#endif
                if (gid > END) { ok = false; }
                else if (gid != END) {
                    slot = (unsigned)lq % (unsigned)R;
                    if (p2) { kind = 2; ok = wait_ge(&ctl->cnt1[xcc][slot][0], NP1 * ((unsigned)lq / R + 1), ctl); }
                    else { kind = 1; ok = wait_ge(&ctl->cnt2[xcc][slot][0], NP2 * ((unsigned)lq / R), ctl); }
                }
            }
#ifdef LAB_STATS
            if (kind) { TICK(t_b); tm_wf += t_b - t_a; }
#endif
            if (!ok) kind = 3;
            else if (kind == 0 && (int)q - L >= 0) {
                // nothing to do in this item: leave once the group whose P2 tiles this block carries is past the end
                unsigned g2 = p2 ? gid : wait_nonzero(&ctl->grp[xcc][q - L], ctl) - 1;
                if (g2 >= END) kind = 3;
            }
            s_kind = kind; s_gid = gid; s_tile = tile; s_slot = slot;
        }
        __syncthreads();
        const unsigned kind = s_kind, gid = s_gid, tile = s_tile, slot = s_slot;
        if (kind == 3) break;
#ifdef LAB_STATS
        unsigned long long t_c = 0, t_d = 0;
        TICK(t_c);
#endif
        f2* const rs = myring + (size_t)slot * SLOT;
        if (kind == 1 && TRIP == 1) {
            // ---- 256-point transform over a, tile (group gid, b = tile): rows (a*16 + b), columns 16 gid + ci
            const int ci = tid & 15, ti = tid >> 4;
            const f2* p = in + (size_t)ti * 65536 + (size_t)tile * 4096 + 16 * gid + ci;
            f2 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = p[(size_t)r * (16 * 65536)];
            F::template compute<16, 1, DIR>(v, ti, tw);
            F::template scatter<16, 1>(v, ti, lds + ci * CS);
            __syncthreads();
            const int c2 = tid >> 4, t2 = tid & 15;
            F::template gather<16>(v, t2, lds + c2 * CS);
            F::template compute<16, 16, DIR>(v, t2, tw);
            f2* o = rs + (tile * 16 + c2) * 256 + t2;
#pragma unroll
            for (int r = 0; r < 16; ++r) if (ring_on) o[16 * r] = v[r];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&ctl->cnt1[xcc][slot][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (kind == 2 && TRIP == 1) {
            // ---- 16-point transform over b for ka = 32 tile + 2 kp (+1), column c = 16 gid + ci
            const int kp = tid & 15, ci = tid >> 4;
            const unsigned ka = 32 * tile + 2 * kp, c = 16 * gid + ci;
            const f2* rp = rs + ci * 256 + ka;
            f2 va[16], vb[16];
#ifdef LAB_BUFLOAD
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(rs, SLOT * 8);
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                f4 u = ld_sc1_16(rr, (unsigned)((b * 4096 + ci * 256 + ka) * 8));
                va[b] = f2{u.x, u.y}; vb[b] = f2{u.z, u.w};
            }
#else
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                f4 u = f4{(float)tid, 1.0f, (float)b, 2.0f};
                if (ring_on) u = ld_sc1_f4(rp + b * 4096);
                va[b] = f2{u.x, u.y}; vb[b] = f2{u.z, u.w};
            }
#endif
            // the slot may be rewritten once every reader's loads have landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&ctl->cnt2[xcc][slot][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // output twiddles W_N^(c (ka + 256 kb)) = [W_N^(c ka) S^k0] S^(4 k1), S = W_65536^c, kb = k0 + 4 k1
            const f2 s1 = root24((c * 256u) & 0xffffffu), s2 = cmul(s1, s1), s3 = cmul(s2, s1);
            const f2 s4 = root24((c * 1024u) & 0xffffffu), s8 = cmul(s4, s4), s12 = cmul(s8, s4);
            auto half = [&](f2* v, unsigned k) {
                const f2 w = root24(k * 4096u); // W_4096^ka
                const f2 held[2] = {cmul(w, w), w};
                f2 t8[8];
                expand_twiddles16_fma<2>(held, t8);
                dft16_tw<DIR>(v, t8);
                f2 b1[4];
                b1[0] = root24((c * k) & 0xffffffu);
                b1[1] = cmul(b1[0], s1); b1[2] = cmul(b1[0], s2); b1[3] = cmul(b1[0], s3);
#pragma unroll
                for (int k0 = 0; k0 < 4; ++k0) {
                    v[k0] = twmul<DIR>(v[k0], b1[k0]);
                    v[4 + k0] = twmul<DIR>(v[4 + k0], cmul(b1[k0], s4));
                    v[8 + k0] = twmul<DIR>(v[8 + k0], cmul(b1[k0], s8));
                    v[12 + k0] = twmul<DIR>(v[12 + k0], cmul(b1[k0], s12));
                }
            };
            half(va, ka);
            half(vb, ka + 1);
            f4* o = reinterpret_cast<f4*>(out + ((size_t)(ka >> 4) * 4096 + c) * 16 + (ka & 15));
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) o[(size_t)kb * (16 * 4096 * 16 / 2)] = f4{va[kb].x, va[kb].y, vb[kb].x, vb[kb].y};
        } else if (kind == 1 && TRIP == 2) {
            // ---- 16-point transform over d for e = 32 tile + ei, kr = 16 gid + 2 kp (+1)
            const int kp = tid & 7, ei = tid >> 3;
            const unsigned e = 32 * tile + ei;
            const f4* p = reinterpret_cast<const f4*>(in + ((size_t)gid * 4096 + e) * 16 + 2 * kp);
            f2 va[16], vb[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                f4 u = p[(size_t)d * (256 * 16 / 2)];
                va[d] = f2{u.x, u.y}; vb[d] = f2{u.z, u.w};
            }
            dft16<DIR>(va);
            dft16<DIR>(vb);
            f2 qp[16];
            powers16(root24(e * 4096u), qp); // W_4096^(e kd)
#pragma unroll
            for (int kd = 1; kd < 16; ++kd) { va[kd] = twmul<DIR>(va[kd], qp[kd]); vb[kd] = twmul<DIR>(vb[kd], qp[kd]); }
            f4* o = reinterpret_cast<f4*>(rs + e * 16 + 2 * kp);
#pragma unroll
            for (int kd = 0; kd < 16; ++kd) if (ring_on) o[kd * (4096 / 2)] = f4{va[kd].x, va[kd].y, vb[kd].x, vb[kd].y};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&ctl->cnt1[xcc][slot][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (kind == 2 && TRIP == 2) {
            // ---- 256-point transform over e, tile (group gid, kd = tile), columns kr%16
            const int c = tid & 15, ti = tid >> 4;
            const f2* rp = rs + tile * 4096 + ti * 16 + c;
            f2 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { v[r] = f2{(float)tid, (float)r}; if (ring_on) v[r] = ld_sc1(rp + r * 256); }
            F::template compute<16, 1, DIR>(v, ti, tw);
            f2* l = lds + c * CS;
            F::template scatter<16, 1>(v, ti, l);
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&ctl->cnt2[xcc][slot][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            F::template gather<16>(v, ti, l);
            F::template compute<16, 16, DIR>(v, ti, tw);
            f2* o = out + 16 * gid + c + (size_t)4096 * tile + (size_t)65536 * ti;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)r * (16 * 65536)] = v[r];
            __syncthreads();
        }
#ifdef LAB_STATS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(t_d);
        if (kind == 1) { tm_p1 += t_d - t_c; ++n_p1; }
        if (kind == 2) { tm_p2 += t_d - t_c; ++n_p2; }
#endif
    }
#ifdef LAB_STATS
    if (tid == 0) {
        const unsigned long long t_end = wall_clock64();
        if (rec) rec[blockIdx.x] = WgRec{t_start, t_end, tm_wf, tm_draw, tm_wg, tm_p1, tm_p2, xcc, (unsigned)n_p1, (unsigned)n_p2, 0};
    }
#endif
    // the last workgroup out resets the control block for the next launch
    __shared__ unsigned s_last;
    if (tid == 0) s_last = __hip_atomic_fetch_add(&ctl->exit_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (s_last) {
        unsigned* w = reinterpret_cast<unsigned*>(ctl);
        const unsigned words = offsetof(Ctl, error) / 4;
        for (unsigned i = tid; i < words; i += 256) __hip_atomic_store(&w[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------ host
static void host_fft(std::vector<double>& re, std::vector<double>& im)
{
    const size_t n = re.size();
    int bits = 0;
    while ((size_t(1) << bits) < n) ++bits;
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0;
        for (int b = 0; b < bits; ++b) j |= ((i >> b) & 1) << (bits - 1 - b);
        if (j > i) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2;
        std::vector<double> wr(half), wi(half);
        for (size_t k = 0; k < half; ++k) { wr[k] = cos(-2.0 * M_PI * k / len); wi[k] = sin(-2.0 * M_PI * k / len); }
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; ++k) {
                const double xr = re[i + k + half] * wr[k] - im[i + k + half] * wi[k];
                const double xi = re[i + k + half] * wi[k] + im[i + k + half] * wr[k];
                re[i + k + half] = re[i + k] - xr; im[i + k + half] = im[i + k] - xi;
                re[i + k] += xr; im[i + k] += xi;
            }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    int per_cu = argc > 1 ? atoi(argv[1]) : 4, L = argc > 2 ? atoi(argv[2]) : 2, R = argc > 3 ? atoi(argv[3]) : 4;
    const bool check = !(argc > 4 && atoi(argv[4]) == 0);
    if (L < 1 || R <= L || R > MAXR) { printf("need 1 <= L < R <= %d\n", MAXR); return 1; }
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t n = NPTS;
    std::vector<f2> h(n);
    unsigned long long st = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        h[i] = f2{(float)((st & 0xffff) / 65536.0 * 20.0 - 10.0), (float)(((st >> 16) & 0xffff) / 65536.0 * 20.0 - 10.0)};
    }
    f2 *x, *mid, *X, *ring, *wt, *lib, *libs;
    Ctl* ctl;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&mid, n * 8)); CK(hipMalloc(&X, n * 8));
    CK(hipMalloc(&lib, n * 8)); CK(hipMalloc(&libs, n * 8));
    CK(hipMalloc(&ring, (size_t)8 * MAXR * SLOT * 8));
    CK(hipMalloc(&wt, 256 * 8));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    CK(hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<f2> hw(256);
    for (int m = 0; m < 256; ++m) hw[m] = f2{(float)cos(-2.0 * M_PI * m / 256), (float)sin(-2.0 * M_PI * m / 256)};
    CK(hipMemcpy(wt, hw.data(), 256 * 8, hipMemcpyHostToDevice));
    const size_t ldsb = (size_t)(16 * CS + 256) * 8;
    const unsigned grid = (unsigned)(cus * per_cu);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run = [&]() {
        hipLaunchKernelGGL((k_trip<1, -1>), dim3(grid), dim3(256), ldsb, s, x, mid, ring, ctl, wt, L, R);
        hipLaunchKernelGGL((k_trip<2, -1>), dim3(grid), dim3(256), ldsb, s, mid, X, ring, ctl, wt, L, R);
    };
    printf("# two-trip 2^24-point f32 FFT: %d CUs x %d workgroups, lag %d, ring %d slots/XCD (%.1f MB per XCD), %d waves/SIMD build\n", cus, per_cu, L, R,
           R * 0.5, LAB_WAVES);
    Ctl hc;
#ifdef LAB_STATS
    for (int rep = 0; rep < 3; ++rep)
        for (int trip = 1; trip <= 2; ++trip) {
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
            { unsigned long long big = ~0ull; CK(hipMemcpyAsync(&ctl->span[0], &big, 8, hipMemcpyHostToDevice, s)); }
            CK(hipStreamSynchronize(s));
            static WgRec* rec = nullptr;
            if (!rec) CK(hipMalloc(&rec, sizeof(WgRec) * grid));
            if (trip == 1) hipLaunchKernelGGL((k_trip<1, -1>), dim3(grid), dim3(256), ldsb, s, x, mid, ring, ctl, wt, L, R, rec);
            else hipLaunchKernelGGL((k_trip<2, -1>), dim3(grid), dim3(256), ldsb, s, mid, X, ring, ctl, wt, L, R, rec);
            CK(hipStreamSynchronize(s));
            {
                std::vector<WgRec> hr(grid);
                CK(hipMemcpy(hr.data(), rec, sizeof(WgRec) * grid, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t1 = 0;
                double a = 0, d = 0, g = 0, f = 0, p1 = 0, p2 = 0, n1 = 0, n2 = 0;
                for (auto& r : hr) {
                    t0 = r.t0 < t0 ? r.t0 : t0; t1 = r.t1 > t1 ? r.t1 : t1;
                    a += (r.t1 - r.t0) * 0.01; d += r.draw * 0.01; g += r.wg * 0.01; f += r.wf * 0.01; p1 += r.p1 * 0.01; p2 += r.p2 * 0.01; n1 += r.np1; n2 += r.np2;
                }
                const double wg = grid;
                printf("  trip %d: span %.1f us; per workgroup: alive %.1f us = draw %.1f + wait-group %.1f + wait-flag %.1f + P1 %.1f (%.1f items, %.2f us each) + P2 %.1f (%.1f items, %.2f us each)\n",
                       trip, (t1 - t0) * 0.01, a / wg, d / wg, g / wg, f / wg, p1 / wg, n1 / wg, n1 ? p1 / n1 : 0.0, p2 / wg, n2 / wg, n2 ? p2 / n2 : 0.0);
                if (rep == 2)
                for (unsigned xc = 0; xc < 8; ++xc) {
                    double n = 0, smax = 0, emin = 1e9, emax = 0, eavg = 0, i1 = 0, i2 = 0, wf = 0;
                    for (auto& r : hr) if (r.xcc == xc) {
                        const double st = (r.t0 - t0) * 0.01, en = (r.t1 - t0) * 0.01;
                        n += 1; smax = st > smax ? st : smax; emin = en < emin ? en : emin; emax = en > emax ? en : emax; eavg += en; i1 += r.np1; i2 += r.np2; wf += r.wf * 0.01;
                    }
                    printf("          XCD %u: %3.0f workgroups, last start %.1f us, exits %.1f .. %.1f (avg %.1f) us, items %4.0f + %4.0f, wait-flag avg %.1f us\n", xc, n, smax, emin, emax,
                           n ? eavg / n : 0, i1, i2, n ? wf / n : 0);
                }
            }
        }
    CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
    CK(hipStreamSynchronize(s));
#endif
    run();
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("first run: error=%u maxspin=%u groups/XCD (both trips)=", hc.error, hc.maxspin);
    for (int i = 0; i < 8; ++i) printf("%u ", hc.stats[i]);
    printf("\n");
    if (check) {
        std::vector<f2> ho(n);
        CK(hipMemcpy(ho.data(), X, n * 8, hipMemcpyDeviceToHost));
        std::vector<double> re(n), im(n);
        for (size_t i = 0; i < n; ++i) { re[i] = h[i].x; im[i] = h[i].y; }
        host_fft(re, im);
        double num = 0, den = 0, mx = 0;
        size_t worst = 0;
        for (size_t i = 0; i < n; ++i) {
            const double dr = ho[i].x - re[i], di = ho[i].y - im[i], e2 = dr * dr + di * di;
            num += e2; den += re[i] * re[i] + im[i] * im[i];
            if (e2 > mx) { mx = e2; worst = i; }
        }
        printf("check vs f64 host transform: rel-L2 %.3e, worst bin %zu abs err %.3e (|X| rms %.3e)\n", sqrt(num / den), worst, sqrt(mx), sqrt(den / n));
        // the library's three-pass transform for the same input
        CK(hipMemcpy(lib, x, n * 8, hipMemcpyDeviceToDevice));
        int in_scratch = 0;
        if (bdsp_hip_dev_fft(0, lib, libs, n, 1, 0, 1.0, -1, 0.0, &in_scratch, s) == 0) {
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(ho.data(), in_scratch ? libs : lib, n * 8, hipMemcpyDeviceToHost));
            num = 0;
            for (size_t i = 0; i < n; ++i) { const double dr = ho[i].x - re[i], di = ho[i].y - im[i]; num += dr * dr + di * di; }
            printf("library three-pass transform: rel-L2 %.3e\n", sqrt(num / den));
        } else printf("library call failed\n");
    }
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int i = 0; i < 5; ++i) run();
    CK(hipStreamSynchronize(s));
    const int iters = 20;
    float t1 = 0, t2 = 0;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL((k_trip<1, -1>), dim3(grid), dim3(256), ldsb, s, x, mid, ring, ctl, wt, L, R);
        CK(hipEventRecord(e1, s));
        hipLaunchKernelGGL((k_trip<2, -1>), dim3(grid), dim3(256), ldsb, s, mid, X, ring, ctl, wt, L, R);
        CK(hipEventRecord(e2, s));
        CK(hipStreamSynchronize(s));
        float a, b;
        CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        t1 += a; t2 += b;
    }
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) run();
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float tb;
    CK(hipEventElapsedTime(&tb, e0, e1));
    CK(hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("trip 1 %.1f us, trip 2 %.1f us (event pairs); back to back %.1f us per transform; error=%u\n", t1 / iters * 1e3, t2 / iters * 1e3,
           tb / iters * 1e3, hc.error);
    // library baseline in the same process
    {
        int in_scratch = 0;
        for (int i = 0; i < 5; ++i) bdsp_hip_dev_fft(0, lib, libs, n, 1, 0, 1.0, -1, 0.0, &in_scratch, s);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) bdsp_hip_dev_fft(0, lib, libs, n, 1, 0, 1.0, -1, 0.0, &in_scratch, s);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&tb, e0, e1));
        printf("library three passes: %.1f us per transform\n", tb / iters * 1e3);
    }
    return 0;
}
