// fft_core.h -- workgroup-level Stockham FFT building blocks for gfx950.
//
// Everything here is a host+device template so the index math can be exercised on the CPU
// (tests/host_sim) where there is no GPU; the kernels in fft_kernels.hip / conv_kernels.hip
// call exactly these functions with threadIdx-derived thread ids and LDS pointers.
//
// Algorithm: autosort Stockham, one iteration per radix-R stage (R in {2,4,8,16}):
//     thread j:  k = j mod Ns
//                v[r] = in[j + r*N/R] * w_{Ns*R}^{r*k}        r = 0..R-1
//                v    = DFT_R(v)
//                out[(j/Ns)*Ns*R + k + r*Ns] = v[r]
// Ns = product of the radices already applied.  Output is in natural order, so there is no
// bit-reversal pass.  A thread owns E = N/NT elements in registers; data crosses threads only
// through LDS between stages (one write + one read per stage boundary).
//
// What this replaces in the reference: the rustfft plan executed by fft()
// (vector/src/vector_types/time_freq/mod.rs:32-63) and the clFFT plans of the OpenCL backend
// (vector/src/gpu_support/ocl/mod.rs:335-349, 395-413).  Unnormalised in both directions.
#pragma once

#if defined(__HIPCC__)
#define BDSP_HD __host__ __device__ __forceinline__
#else
#define BDSP_HD inline
#endif

namespace bdsp {

template <typename T>
struct alignas(2 * sizeof(T)) cpx {
    T x, y;
};

template <typename T>
BDSP_HD cpx<T> cadd(cpx<T> a, cpx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
BDSP_HD cpx<T> csub(cpx<T> a, cpx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
BDSP_HD cpx<T> cmul(cpx<T> a, cpx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * conj(b)
template <typename T>
BDSP_HD cpx<T> cmulc(cpx<T> a, cpx<T> b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
// DIR = -1: forward (multiply by -i), DIR = +1: inverse (multiply by +i)
template <int DIR, typename T>
BDSP_HD cpx<T> mul_dir_i(cpx<T> a) { return DIR < 0 ? cpx<T>{a.y, -a.x} : cpx<T>{-a.y, a.x}; }
// twiddle application: forward uses w, inverse uses conj(w); tables always hold forward values
template <int DIR, typename T>
BDSP_HD cpx<T> twmul(cpx<T> a, cpx<T> w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }

// ------------------------------------------------------------------ small DFTs, natural order
template <int DIR, typename T>
BDSP_HD void dft2(cpx<T>& a, cpx<T>& b)
{
    cpx<T> t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

template <int DIR, typename T>
BDSP_HD void dft4(cpx<T>& a0, cpx<T>& a1, cpx<T>& a2, cpx<T>& a3)
{
    cpx<T> t0 = cadd(a0, a2), t1 = csub(a0, a2);
    cpx<T> t2 = cadd(a1, a3), t3 = mul_dir_i<DIR>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

// a * exp(DIR * i*pi/4) and a * exp(DIR * 3i*pi/4)
template <int DIR, typename T>
BDSP_HD cpx<T> mul_w8_1(cpx<T> a)
{
    const T h = (T)0.70710678118654752440;
    return DIR < 0 ? cpx<T>{(a.x + a.y) * h, (a.y - a.x) * h} : cpx<T>{(a.x - a.y) * h, (a.x + a.y) * h};
}
template <int DIR, typename T>
BDSP_HD cpx<T> mul_w8_3(cpx<T> a)
{
    const T h = (T)0.70710678118654752440;
    return DIR < 0 ? cpx<T>{(a.y - a.x) * h, -(a.x + a.y) * h} : cpx<T>{-(a.x + a.y) * h, (a.x - a.y) * h};
}

template <int DIR, typename T>
BDSP_HD void dft8(cpx<T>* v)
{
    // even / odd split
    dft4<DIR>(v[0], v[2], v[4], v[6]); // E[k] in v[0],v[2],v[4],v[6]
    dft4<DIR>(v[1], v[3], v[5], v[7]); // O[k] in v[1],v[3],v[5],v[7]
    cpx<T> o1 = mul_w8_1<DIR>(v[3]);
    cpx<T> o2 = mul_dir_i<DIR>(v[5]);
    cpx<T> o3 = mul_w8_3<DIR>(v[7]);
    cpx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

template <int DIR, typename T>
BDSP_HD void dft16(cpx<T>* v)
{
    // n = 4*n1 + n2, k = k1 + 4*k2.  Step 1: DFT4 over n1 for each n2 -> B[n2][k1] left in
    // v[4*k1 + n2]; step 2: * w16^(n2*k1); step 3: DFT4 over n2 for each k1 -> X[k1 + 4*k2].
    dft4<DIR>(v[0], v[4], v[8], v[12]);
    dft4<DIR>(v[1], v[5], v[9], v[13]);
    dft4<DIR>(v[2], v[6], v[10], v[14]);
    dft4<DIR>(v[3], v[7], v[11], v[15]);
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173; // cos/sin(pi/8)
    const cpx<T> w1 = {c1, -s1}, w3 = {s1, -c1};                             // forward w16^1, w16^3
    // k1 = 1 : v[4+n2] *= w16^(n2)
    v[5] = twmul<DIR>(v[5], w1);
    v[6] = mul_w8_1<DIR>(v[6]);
    v[7] = twmul<DIR>(v[7], w3);
    // k1 = 2 : v[8+n2] *= w16^(2 n2) = w8^(n2)
    v[9] = mul_w8_1<DIR>(v[9]);
    v[10] = mul_dir_i<DIR>(v[10]);
    v[11] = mul_w8_3<DIR>(v[11]);
    // k1 = 3 : v[12+n2] *= w16^(3 n2): n2=1 -> w16^3, n2=2 -> w16^6 = w8^3, n2=3 -> w16^9 = -w16^1
    v[13] = twmul<DIR>(v[13], w3);
    v[14] = mul_w8_3<DIR>(v[14]);
    {
        cpx<T> t = twmul<DIR>(v[15], w1);
        v[15] = cpx<T>{-t.x, -t.y};
    }
    dft4<DIR>(v[0], v[1], v[2], v[3]);     // k1 = 0 -> X[0], X[4], X[8], X[12]
    dft4<DIR>(v[4], v[5], v[6], v[7]);     // k1 = 1 -> X[1], X[5], X[9], X[13]
    dft4<DIR>(v[8], v[9], v[10], v[11]);   // k1 = 2
    dft4<DIR>(v[12], v[13], v[14], v[15]); // k1 = 3
    // v[4*k1 + k2] holds X[k1 + 4*k2]: transpose the 4x4 to natural order
    cpx<T> t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int R, int DIR, typename T>
BDSP_HD void dft(cpx<T>* v)
{
    if (R == 2) dft2<DIR>(v[0], v[1]);
    else if (R == 4) dft4<DIR>(v[0], v[1], v[2], v[3]);
    else if (R == 8) dft8<DIR>(v);
    else if (R == 16) dft16<DIR>(v);
}

// ------------------------------------------------------------------ workgroup FFT
// N points, NT cooperating threads, E = N/NT elements per thread (E in {2,4,8,16}, E >= every
// radix used).  LDS holds N elements padded by one element per 16 so the stride-R scatter of the
// first stage and the contiguous gathers are (nearly) bank-conflict free for ds_*_b64/b128.
template <typename T, int N, int NT>
struct WgFft {
    static constexpr int E = N / NT;
    static constexpr int LDS_ELEMS = N + (N >> 4);
    static BDSP_HD int pad(int i) { return i + (i >> 4); }

    // Register <-> data index convention for a radix-R stage: butterfly b (0..E/R-1) of thread t
    // works on column index j = t + b*NT and owns v[b*R + r] <-> data[j + r*N/R].
    template <int R>
    static BDSP_HD int in_index(int t, int b, int r) { return t + b * NT + r * (N / R); }
    // natural-order output index of the same register after a stage with previous product NS
    template <int R, int NS>
    static BDSP_HD int out_index(int t, int b, int r)
    {
        int j = t + b * NT;
        return (j / NS) * NS * R + (j % NS) + r * NS;
    }

    // twiddle + butterfly.  `tw(m)` returns the FORWARD table value exp(-2*pi*i*m/N), m in [0,N).
    template <int R, int NS, int DIR, class TW>
    static BDSP_HD void compute(cpx<T> (&v)[E], int t, TW tw)
    {
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            if (NS > 1) {
                int k = (t + b * NT) % NS;
#pragma unroll
                for (int r = 1; r < R; ++r)
                    v[b * R + r] = twmul<DIR>(v[b * R + r], tw(r * k * (N / (NS * R))));
            }
            dft<R, DIR>(&v[b * R]);
        }
    }

    // same, twiddles handed in as a per-thread register array tws[b*(R-1) + r-1]
    template <int R, int NS, int DIR>
    static BDSP_HD void compute_pre(cpx<T> (&v)[E], const cpx<T>* tws)
    {
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
#pragma unroll
            for (int r = 1; r < R; ++r)
                v[b * R + r] = twmul<DIR>(v[b * R + r], tws[b * (R - 1) + r - 1]);
            dft<R, DIR>(&v[b * R]);
        }
    }
    template <int R, int NS, class TW>
    static BDSP_HD void load_twiddles(cpx<T>* tws, int t, TW tw)
    {
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            int k = (t + b * NT) % NS;
#pragma unroll
            for (int r = 1; r < R; ++r) tws[b * (R - 1) + r - 1] = tw(r * k * (N / (NS * R)));
        }
    }

    // LDS addressing is kept in the form  base(thread) + constant(register)  so the constants fold
    // into the 16-bit immediate offset of ds_read/ds_write and no per-register address VGPRs are
    // held across the persistent block loop:
    //   pad(i + 16*c) = pad(i) + 17*c, and for the first stage (NS = 1, R = 16) pad(16*j + r) = 17*j + r.
    template <int R, int NS>
    static BDSP_HD void scatter(const cpx<T> (&v)[E], int t, cpx<T>* lds)
    {
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            const int j = t + b * NT;
            if (NS % 16 == 0) {
                cpx<T>* p = lds + pad((j / NS) * NS * R + (j % NS));
#pragma unroll
                for (int r = 0; r < R; ++r) p[r * (NS / 16) * 17] = v[b * R + r];
            } else if (NS == 1 && R == 16) {
                cpx<T>* p = lds + 17 * j;
#pragma unroll
                for (int r = 0; r < R; ++r) p[r] = v[b * R + r];
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) lds[pad(out_index<R, NS>(t, b, r))] = v[b * R + r];
            }
        }
    }

    template <int R>
    static BDSP_HD void gather(cpx<T> (&v)[E], int t, const cpx<T>* lds)
    {
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            if ((N / R) % 16 == 0) {
                const cpx<T>* p = lds + pad(t + b * NT);
#pragma unroll
                for (int r = 0; r < R; ++r) v[b * R + r] = p[r * ((N / R) / 16) * 17];
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) v[b * R + r] = lds[pad(in_index<R>(t, b, r))];
            }
        }
    }
};

// Radix plan for an N-point workgroup FFT with E = 16 registers per thread (N >= 16):
// up to three stages R1*R2*R3 = N, all 16 except the last.
template <int N>
struct Radix16Plan {
    static constexpr int R1 = 16;
    static constexpr int R2 = (N / 16) >= 16 ? 16 : (N / 16);          // 1 if N == 16
    static constexpr int R3 = (N / 16 / R2);                            // 1 if N <= 256
    static_assert(R1 * R2 * R3 == N, "N must be 16 * 2^k <= 4096");
};

} // namespace bdsp
