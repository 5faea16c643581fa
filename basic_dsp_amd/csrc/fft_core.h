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

// Complex value type.  f64 (and every host build) uses a plain struct.  On the device, f32 complex
// values are 2-wide clang vectors so that complex add/sub/multiply lower to gfx950's PACKED f32
// VALU ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, with the swaps and sign flips of the
// multiply-by-i and conjugate forms folded into op_sel / neg modifiers).  Measured on MI355X
// (tools/ubench/valu_rate.hip): one wave issues a VALU instruction about every 5-6 clocks whatever
// its width, so a lone wave on a SIMD retires 2x the flops with packed ops; v_pk_fma_f32 sustains
// 2.8 clk per result with two waves, scalar v_fma_f32 4.6.
template <typename T>
struct alignas(2 * sizeof(T)) cpx_s {
    T x, y;
};
template <typename C> struct real_of;
template <typename T> struct real_of<cpx_s<T>> { using type = T; };

#if defined(__HIP_DEVICE_COMPILE__) || (defined(__HIPCC__) && defined(__clang__))
typedef float bdsp_f32x2 __attribute__((ext_vector_type(2)));
template <> struct real_of<bdsp_f32x2> { using type = float; };
template <typename T> struct cpx_sel { using type = cpx_s<T>; };
template <> struct cpx_sel<float> { using type = bdsp_f32x2; };
#define BDSP_PACKED_F32 1
#else
template <typename T> struct cpx_sel { using type = cpx_s<T>; };
#endif
template <typename T> using cpx = typename cpx_sel<T>::type;

template <typename C> BDSP_HD C cadd(C a, C b) { return C{a.x + b.x, a.y + b.y}; }
template <typename C> BDSP_HD C csub(C a, C b) { return C{a.x - b.x, a.y - b.y}; }
template <typename C> BDSP_HD C cmul(C a, C b) { return C{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * conj(b)
template <typename C> BDSP_HD C cmulc(C a, C b) { return C{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
template <typename C> BDSP_HD C cscale(C a, typename real_of<C>::type s) { return C{a.x * s, a.y * s}; }
// DIR = -1: forward (multiply by -i), DIR = +1: inverse (multiply by +i)
template <int DIR, typename C>
BDSP_HD C mul_dir_i(C a) { return DIR < 0 ? C{a.y, -a.x} : C{-a.y, a.x}; }

// a + (DIR*i)*d and a - (DIR*i)*d (the rotated sums of a radix-4 butterfly)
template <int DIR, typename C>
BDSP_HD C cadd_i(C a, C d) { return cadd(a, mul_dir_i<DIR>(d)); }
template <int DIR, typename C>
BDSP_HD C csub_i(C a, C d) { return csub(a, mul_dir_i<DIR>(d)); }

#ifdef BDSP_PACKED_F32
// Packed f32 forms (VOP3P).  op_sel[i] / op_sel_hi[i] pick the half of source i that feeds the
// low / high result lane, neg_lo / neg_hi negate it: every swap, broadcast and sign flip of
// complex arithmetic rides on the instruction as a modifier, so nothing is materialised in extra
// registers (hipcc's own selection spent 124 v_xor + 315 v_mov per block and doubled the twiddle
// registers).  Plain VALU read-after-write is interlocked in hardware: no manual wait states.
BDSP_HD bdsp_f32x2 cadd(bdsp_f32x2 a, bdsp_f32x2 b) { return a + b; }
BDSP_HD bdsp_f32x2 csub(bdsp_f32x2 a, bdsp_f32x2 b) { return a - b; }
BDSP_HD bdsp_f32x2 cscale(bdsp_f32x2 a, float s) { return a * bdsp_f32x2{s, s}; }
// One asm statement per INSTRUCTION, deliberately: merging the mul+fma of a complex multiply (or a
// whole radix-4 butterfly) into one statement measured 8-12 % slower on the overlap-save kernel --
// the statements are opaque to the scheduler, so the dependent instructions inside issue back to
// back and stall, while separate statements let hipcc interleave independent butterflies.
BDSP_HD bdsp_f32x2 cmul(bdsp_f32x2 a, bdsp_f32x2 b)
{
    bdsp_f32x2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b)); // (ax bx, ax by)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"        // (-ay by, ay bx) + t
        : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
BDSP_HD bdsp_f32x2 cmulc(bdsp_f32x2 a, bdsp_f32x2 b)
{
    bdsp_f32x2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b)); // (ax bx, -ax by)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"                                    // (ay by, ay bx) + t
        : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// forward (-i d) = (dy, -dx); inverse (+i d) = (-dy, dx)
template <int DIR>
BDSP_HD bdsp_f32x2 cadd_i(bdsp_f32x2 a, bdsp_f32x2 d)
{
    bdsp_f32x2 r;
    if (DIR < 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
template <int DIR>
BDSP_HD bdsp_f32x2 csub_i(bdsp_f32x2 a, bdsp_f32x2 d) { return cadd_i<-DIR>(a, d); }
template <int DIR>
BDSP_HD bdsp_f32x2 mul_dir_i_pk(bdsp_f32x2 a)
{
    bdsp_f32x2 r;
    if (DIR < 0) asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
    else asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
#endif

// standalone multiply by -/+ i
template <int DIR, typename C>
BDSP_HD C rot_i(C a) { return mul_dir_i<DIR>(a); }
#ifdef BDSP_PACKED_F32
template <int DIR>
BDSP_HD bdsp_f32x2 rot_i(bdsp_f32x2 a) { return mul_dir_i_pk<DIR>(a); }
#endif

// twiddle application: forward uses w, inverse uses conj(w); tables always hold forward values
template <int DIR, typename C>
BDSP_HD C twmul(C a, C w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }

// ------------------------------------------------------------------ small DFTs, natural order
template <int DIR, typename C>
BDSP_HD void dft2(C& a, C& b)
{
    C t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

template <int DIR, typename C>
BDSP_HD void dft4(C& a0, C& a1, C& a2, C& a3)
{
    C t0 = cadd(a0, a2), t1 = csub(a0, a2);
    C t2 = cadd(a1, a3), d = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd_i<DIR>(t1, d);
    a3 = csub_i<DIR>(t1, d);
}

// a * exp(DIR * i*pi/4) and a * exp(DIR * 3i*pi/4):  (a + (-/+ i) a) * h  and  (-a + (-/+ i) a) * h... spelled
// so that each is one packed add (with a swizzled, sign-flipped second operand) and one packed mul
template <int DIR, typename C>
BDSP_HD C mul_w8_1(C a)
{
    using T = typename real_of<C>::type;
    const T h = (T)0.70710678118654752440;
    // forward: (ax + ay, ay - ax) h = (a + (ay, -ax)) h ; inverse: (ax - ay, ax + ay) h = (a + (-ay, ax)) h
    return cscale(cadd_i<DIR>(a, a), h);
}
template <int DIR, typename C>
BDSP_HD C mul_w8_3(C a)
{
    using T = typename real_of<C>::type;
    const T h = (T)0.70710678118654752440;
    // forward: (ay - ax, -(ax + ay)) h = -(a - (ay, -ax)) h ; inverse: (-(ax + ay), ax - ay) h = -(a - (-ay, ax)) h
    return cscale(csub_i<DIR>(a, a), -h);
}

template <int DIR, typename C>
BDSP_HD void dft8(C* v)
{
    // even / odd split
    dft4<DIR>(v[0], v[2], v[4], v[6]); // E[k] in v[0],v[2],v[4],v[6]
    dft4<DIR>(v[1], v[3], v[5], v[7]); // O[k] in v[1],v[3],v[5],v[7]
    C o1 = mul_w8_1<DIR>(v[3]);
    C o2 = rot_i<DIR>(v[5]);
    C o3 = mul_w8_3<DIR>(v[7]);
    C e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

// dft16 in two halves (dft16 = dft16_a; dft16_b) for kernels that put a barrier in the middle
template <int DIR, typename C>
BDSP_HD void dft16_a(C* v)
{
    using T = typename real_of<C>::type;
    dft4<DIR>(v[0], v[4], v[8], v[12]);
    dft4<DIR>(v[1], v[5], v[9], v[13]);
    dft4<DIR>(v[2], v[6], v[10], v[14]);
    dft4<DIR>(v[3], v[7], v[11], v[15]);
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;
    const C w1 = {c1, -s1}, w3 = {s1, -c1};
    v[5] = twmul<DIR>(v[5], w1);
    v[6] = mul_w8_1<DIR>(v[6]);
    v[7] = twmul<DIR>(v[7], w3);
    v[9] = mul_w8_1<DIR>(v[9]);
    v[10] = rot_i<DIR>(v[10]);
    v[11] = mul_w8_3<DIR>(v[11]);
    v[13] = twmul<DIR>(v[13], w3);
    v[14] = mul_w8_3<DIR>(v[14]);
    v[15] = twmul<DIR>(v[15], C{-c1, s1});
}
template <int DIR, typename C>
BDSP_HD void dft16_b(C* v)
{
    dft4<DIR>(v[0], v[1], v[2], v[3]);
    dft4<DIR>(v[4], v[5], v[6], v[7]);
    dft4<DIR>(v[8], v[9], v[10], v[11]);
    dft4<DIR>(v[12], v[13], v[14], v[15]);
    C t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int DIR, typename C>
BDSP_HD void dft16(C* v)
{
    using T = typename real_of<C>::type;
    // n = 4*n1 + n2, k = k1 + 4*k2.  Step 1: DFT4 over n1 for each n2 -> B[n2][k1] left in
    // v[4*k1 + n2]; step 2: * w16^(n2*k1); step 3: DFT4 over n2 for each k1 -> X[k1 + 4*k2].
    dft4<DIR>(v[0], v[4], v[8], v[12]);
    dft4<DIR>(v[1], v[5], v[9], v[13]);
    dft4<DIR>(v[2], v[6], v[10], v[14]);
    dft4<DIR>(v[3], v[7], v[11], v[15]);
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173; // cos/sin(pi/8)
    const C w1 = {c1, -s1}, w3 = {s1, -c1};                                  // forward w16^1, w16^3
    // k1 = 1 : v[4+n2] *= w16^(n2)
    v[5] = twmul<DIR>(v[5], w1);
    v[6] = mul_w8_1<DIR>(v[6]);
    v[7] = twmul<DIR>(v[7], w3);
    // k1 = 2 : v[8+n2] *= w16^(2 n2) = w8^(n2)
    v[9] = mul_w8_1<DIR>(v[9]);
    v[10] = rot_i<DIR>(v[10]);
    v[11] = mul_w8_3<DIR>(v[11]);
    // k1 = 3 : v[12+n2] *= w16^(3 n2): n2=1 -> w16^3, n2=2 -> w16^6 = w8^3, n2=3 -> w16^9 = -w16^1
    v[13] = twmul<DIR>(v[13], w3);
    v[14] = mul_w8_3<DIR>(v[14]);
    v[15] = twmul<DIR>(v[15], C{-c1, s1});
    dft4<DIR>(v[0], v[1], v[2], v[3]);     // k1 = 0 -> X[0], X[4], X[8], X[12]
    dft4<DIR>(v[4], v[5], v[6], v[7]);     // k1 = 1 -> X[1], X[5], X[9], X[13]
    dft4<DIR>(v[8], v[9], v[10], v[11]);   // k1 = 2
    dft4<DIR>(v[12], v[13], v[14], v[15]); // k1 = 3
    // v[4*k1 + k2] holds X[k1 + 4*k2]: transpose the 4x4 to natural order
    C t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

#ifdef BDSP_PACKED_F32
// c + a * s (s a real scalar) and c + s * (DIR*i) * w: one packed fma each; the rotation by +-i is an operand
// swizzle with a sign flip on one half
BDSP_HD bdsp_f32x2 fma_s(bdsp_f32x2 a, float s, bdsp_f32x2 c)
{
    bdsp_f32x2 r, ss = {s, s};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(ss), "v"(c));
    return r;
}
template <int DIR>
BDSP_HD bdsp_f32x2 fma_rot(bdsp_f32x2 w, float s, bdsp_f32x2 c)
{
    bdsp_f32x2 r, ss = {s, s};
    // forward: (w.y, -w.x) * s + c ; inverse: (-w.y, w.x) * s + c
    if (DIR < 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(ss), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(w), "v"(ss), "v"(c));
    return r;
}
// The 16-point transform with the scalings by sqrt(1/2) of its w8 twiddles folded into the multiply-adds of the second
// butterfly layer (and the lone rotation by -+i into an add): 76 packed instructions instead of 81.  The overlap-save
// block kernel is bound by its instruction count, six of these per block.
template <int DIR>
BDSP_HD void dft16(bdsp_f32x2* v)
{
    using C = bdsp_f32x2;
    const float h = 0.70710678118654752440f;
    dft4<DIR>(v[0], v[4], v[8], v[12]);
    dft4<DIR>(v[1], v[5], v[9], v[13]);
    dft4<DIR>(v[2], v[6], v[10], v[14]);
    dft4<DIR>(v[3], v[7], v[11], v[15]);
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
    const C w1 = {c1, -s1}, w3 = {s1, -c1};
    // row k1 = 0
    dft4<DIR>(v[0], v[1], v[2], v[3]);
    // row k1 = 1: (v4, v5 w16, v6 w8, v7 w16^3); v6 w8 = h (v6 + (DIR i) v6)
    {
        const C a1 = twmul<DIR>(v[5], w1), a3 = twmul<DIR>(v[7], w3), s6 = cadd_i<DIR>(v[6], v[6]);
        const C t0 = fma_s(s6, h, v[4]), t1 = fma_s(s6, -h, v[4]);
        const C t2 = cadd(a1, a3), d = csub(a1, a3);
        v[4] = cadd(t0, t2); v[6] = csub(t0, t2);
        v[5] = cadd_i<DIR>(t1, d); v[7] = csub_i<DIR>(t1, d);
    }
    // row k1 = 2: (v8, v9 w8, v10 (DIR i), v11 w8^3); v9 w8 = h s9, v11 w8^3 = -h s11
    {
        const C s9 = cadd_i<DIR>(v[9], v[9]), s11 = csub_i<DIR>(v[11], v[11]);
        const C t0 = cadd_i<DIR>(v[8], v[10]), t1 = csub_i<DIR>(v[8], v[10]);
        const C u = csub(s9, s11), w = cadd(s9, s11); // a1 + a3 = h u, a1 - a3 = h w
        v[8] = fma_s(u, h, t0); v[10] = fma_s(u, -h, t0);
        v[9] = fma_rot<DIR>(w, h, t1); v[11] = fma_rot<DIR>(w, -h, t1);
    }
    // row k1 = 3: (v12, v13 w16^3, v14 w8^3, v15 w16^9 = -w16); v14 w8^3 = -h (v14 - (DIR i) v14)
    {
        const C a1 = twmul<DIR>(v[13], w3), a3 = twmul<DIR>(v[15], C{-c1, s1}), s14 = csub_i<DIR>(v[14], v[14]);
        const C t0 = fma_s(s14, -h, v[12]), t1 = fma_s(s14, h, v[12]);
        const C t2 = cadd(a1, a3), d = csub(a1, a3);
        v[12] = cadd(t0, t2); v[14] = csub(t0, t2);
        v[13] = cadd_i<DIR>(t1, d); v[15] = csub_i<DIR>(t1, d);
    }
    C t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}
#endif

// ------------------------------------------------------------------ twiddled 16-point transform, FMA form
// X[q] = sum_r u[r] w^r W16^(r q), i.e. the radix-16 butterfly of a Stockham stage TOGETHER with its fifteen input
// twiddles w^r, as four layers of radix-2 butterflies whose twiddle rides on the multiply-adds:
//     s = a + (rho T) b   (two packed fma: a + Tx b, then + Ty (DIR i) b)        d = 2 a - s   (one fma)
// Layer l (pairs 8 / 4 / 2 / 1 registers apart) needs T = w^8 | w^4 (-i)^q0 | w^2 W8^q0 (-i)^q1 |
// w W16^(q0 + 2 q1) (-i)^q2; the rotations by -+i and the conjugation of the inverse direction are operand modifiers,
// so EIGHT table values serve all 32 butterflies: T[0] = w^8, T[1] = w^4, T[2] = w^2, T[3] = w^2 W8, T[4 + j] = w W16^j.
// 96 packed instructions against 30 (fifteen complex multiplies) + 76 for twiddle-then-dft16, and 8 twiddle values
// per thread instead of 15.  Round 3: the overlap-save block kernel is bound by its instruction count.
// PRUNE > 0 (<= 8): outputs X[0 .. PRUNE) are not needed (the block kernel discards its first rows): their
// butterflies compute only d = a - T b (two fma).
template <int DIR, bool ROT, typename C>
BDSP_HD C bf_tw_s(C a, C b, C T)
{
    C w = DIR < 0 ? T : C{T.x, -T.y};
    if (ROT) w = mul_dir_i<DIR>(w);
    C u = C{a.x + w.x * b.x, a.y + w.x * b.y};
    return C{u.x - w.y * b.y, u.y + w.y * b.x};
}
template <int DIR, bool ROT, typename C>
BDSP_HD C bf_tw_d(C a, C b, C T)
{
    C w = DIR < 0 ? T : C{T.x, -T.y};
    if (ROT) w = mul_dir_i<DIR>(w);
    C u = C{a.x - w.x * b.x, a.y - w.x * b.y};
    return C{u.x + w.y * b.y, u.y - w.y * b.x};
}
template <typename C>
BDSP_HD C bf_2a_minus_s(C a, C s)
{
    using T = typename real_of<C>::type;
    return C{(T)2 * a.x - s.x, (T)2 * a.y - s.y};
}
#ifdef BDSP_PACKED_F32
// u = a +- (Tx | Ty) b, then s = u +- (Ty | Tx) (i b): the four (DIR, ROT) cases differ only in which half of T each
// instruction broadcasts and which half of the swizzled b is negated.
template <int DIR, bool ROT>
BDSP_HD bdsp_f32x2 bf_tw_s(bdsp_f32x2 a, bdsp_f32x2 b, bdsp_f32x2 T)
{
    bdsp_f32x2 u, s;
    if (!ROT) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(u) : "v"(b), "v"(T), "v"(a));
        if (DIR < 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(u) : "v"(b), "v"(T), "v"(a));
        if (DIR < 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
    }
    return s;
}
template <int DIR, bool ROT>
BDSP_HD bdsp_f32x2 bf_tw_d(bdsp_f32x2 a, bdsp_f32x2 b, bdsp_f32x2 T)
{
    bdsp_f32x2 u, s;
    if (!ROT) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(u) : "v"(b), "v"(T), "v"(a));
        if (DIR < 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(u) : "v"(b), "v"(T), "v"(a));
        if (DIR < 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(s) : "v"(b), "v"(T), "v"(u));
    }
    return s;
}
BDSP_HD bdsp_f32x2 bf_2a_minus_s(bdsp_f32x2 a, bdsp_f32x2 s)
{
    bdsp_f32x2 d;
    // (d = a - (s - a) as two packed adds, no multiplier: 60.6 -> 64.3 us on the block kernel -- an instruction is an
    // instruction, whatever it computes)
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(s));
    return d;
}
#endif

template <int DIR, bool ROT, bool NEED_S, typename C>
BDSP_HD void bf_tw(C& a, C& b, C T)
{
    if (NEED_S) {
        const C s = bf_tw_s<DIR, ROT>(a, b, T);
        b = bf_2a_minus_s(a, s);
        a = s;
    } else {
        b = bf_tw_d<DIR, ROT>(a, b, T);
    }
}

template <int DIR, int PRUNE = 0, typename C>
BDSP_HD void dft16_tw(C* v, const C* T)
{
    static_assert(PRUNE >= 0 && PRUNE <= 8, "only the last layer is pruned");
    // layer 1: pairs (r, r + 8), q0 = 0 stays in r, q0 = 1 goes to r + 8
#pragma unroll
    for (int r = 0; r < 8; ++r) bf_tw<DIR, false, true>(v[r], v[r + 8], T[0]);
    // layer 2: pairs (r, r + 4) inside each q0 half; twiddle w^4 (-i)^q0
#pragma unroll
    for (int r = 0; r < 4; ++r) bf_tw<DIR, false, true>(v[r], v[r + 4], T[1]);
#pragma unroll
    for (int r = 0; r < 4; ++r) bf_tw<DIR, true, true>(v[8 + r], v[8 + r + 4], T[1]);
    // layer 3: pairs (r, r + 2) inside each (q0, q1); twiddle w^2 W8^q0 (-i)^q1
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        bf_tw<DIR, false, true>(v[r], v[r + 2], T[2]);
        bf_tw<DIR, true, true>(v[4 + r], v[4 + r + 2], T[2]);
        bf_tw<DIR, false, true>(v[8 + r], v[8 + r + 2], T[3]);
        bf_tw<DIR, true, true>(v[12 + r], v[12 + r + 2], T[3]);
    }
    // layer 4: pairs (2 m, 2 m + 1), m = q2 + 2 q1 + 4 q0; twiddle w W16^(q0 + 2 q1) (-i)^q2; the pair yields
    // X[q] (q = q0 + 2 q1 + 4 q2 < 8) and X[q + 8]
    bf_tw<DIR, false, (0 >= PRUNE)>(v[0], v[1], T[4]);   // q = 0
    bf_tw<DIR, true, (4 >= PRUNE)>(v[2], v[3], T[4]);    // q = 4
    bf_tw<DIR, false, (2 >= PRUNE)>(v[4], v[5], T[6]);   // q = 2
    bf_tw<DIR, true, (6 >= PRUNE)>(v[6], v[7], T[6]);    // q = 6
    bf_tw<DIR, false, (1 >= PRUNE)>(v[8], v[9], T[5]);   // q = 1
    bf_tw<DIR, true, (5 >= PRUNE)>(v[10], v[11], T[5]);  // q = 5
    bf_tw<DIR, false, (3 >= PRUNE)>(v[12], v[13], T[7]); // q = 3
    bf_tw<DIR, true, (7 >= PRUNE)>(v[14], v[15], T[7]);  // q = 7
    // v[bitrev4(q)] holds X[q]
    C t;
    t = v[1]; v[1] = v[8]; v[8] = t;
    t = v[2]; v[2] = v[4]; v[4] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[5]; v[5] = v[10]; v[10] = t;
    t = v[7]; v[7] = v[14]; v[14] = t;
    t = v[11]; v[11] = v[13]; v[13] = t;
}

// The twiddled 8-point transform in the same FMA form: X[q] = sum_r u[r] w^r W8^(r q) as three radix-2 layers, twelve
// butterflies of three instructions; T[0] = w^4, T[1] = w^2, T[2] = w, T[3] = w W8 (four table values instead of seven).
template <int DIR, typename C>
BDSP_HD void dft8_tw(C* v, const C* T)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) bf_tw<DIR, false, true>(v[r], v[r + 4], T[0]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        bf_tw<DIR, false, true>(v[r], v[r + 2], T[1]);
        bf_tw<DIR, true, true>(v[4 + r], v[6 + r], T[1]);
    }
    bf_tw<DIR, false, true>(v[0], v[1], T[2]); // X[0], X[4]
    bf_tw<DIR, true, true>(v[2], v[3], T[2]);  // X[2], X[6]
    bf_tw<DIR, false, true>(v[4], v[5], T[3]); // X[1], X[5]
    bf_tw<DIR, true, true>(v[6], v[7], T[3]);  // X[3], X[7]
    C t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[3]; v[3] = v[6]; v[6] = t;
}

// The eight values of dft16_tw from FOUR held ones q = {w^8, w^4, w^2, w}: the other four are products with the
// constants W8, W16, W16^3 (for kernels short of registers: f64 holds 16 instead of 32 registers of twiddles per stage).
// HELD = 2: q = {w^2, w} only; w^4 and w^8 by squaring (six more multiply-adds).
template <int HELD = 4, typename C>
BDSP_HD void expand_twiddles16_fma(const C* q, C* T)
{
    using R = typename real_of<C>::type;
    const R h = (R)0.70710678118654752440, c1 = (R)0.92387953251128675613, s1 = (R)0.38268343236508977173;
    const C w2 = HELD == 2 ? q[0] : q[2], w1 = HELD == 2 ? q[1] : q[3];
    if (HELD == 2) {
        const C w4 = C{(w2.x - w2.y) * (w2.x + w2.y), (R)2 * w2.x * w2.y};
        T[1] = w4;
        T[0] = C{(w4.x - w4.y) * (w4.x + w4.y), (R)2 * w4.x * w4.y};
    } else {
        T[0] = q[0];
        T[1] = q[1];
    }
    T[2] = w2;
    T[3] = C{(w2.x + w2.y) * h, (w2.y - w2.x) * h}; // * W8 = (h, -h)
    T[4] = w1;
    T[5] = cmul(w1, C{c1, -s1});
    T[6] = C{(w1.x + w1.y) * h, (w1.y - w1.x) * h};
    T[7] = cmul(w1, C{s1, -c1});
}

template <int R, int DIR, typename C>
BDSP_HD void dft(C* v)
{
    if (R == 2) dft2<DIR>(v[0], v[1]);
    else if (R == 4) dft4<DIR>(v[0], v[1], v[2], v[3]);
    else if (R == 8) dft8<DIR>(v);
    else if (R == 16) dft16<DIR>(v);
}

// The real (IM = false) or imaginary parts of a register array of plain-struct complex values, indexable like an array
// of scalars: what WgFft::scatter / gather move when a kernel exchanges the two parts one after the other.
template <typename C, bool IM>
struct PartRef {
    C* v;
    BDSP_HD typename real_of<C>::type& operator[](int i) const { return IM ? v[i].y : v[i].x; }
};

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
        if constexpr (R == 16 && NS > 1 && E == 16 && N % (16 * NS) == 0) {
            // a twiddled radix-16 stage: the FMA form (dft16_tw, round 3) -- eight table values instead of fifteen
            // (they come from L2 or an LDS table in the pass kernels) and 96 instead of 106 packed instructions
            cpx<T> tws[8];
            load_twiddles16_fma<NS>(tws, t, tw);
            dft16_tw<DIR>(&v[0], tws);
            return;
        }
        if constexpr (R == 8 && NS > 1 && N % (8 * NS) == 0) {
#pragma unroll
            for (int b = 0; b < E / R; ++b) {
                const int e = ((t + b * NT) % NS) * (N / (NS * 8));
                const cpx<T> tws[4] = {tw(4 * e), tw(2 * e), tw(e), tw(e + N / 8)};
                dft8_tw<DIR>(&v[b * R], tws);
            }
            return;
        }
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
    // Radix-16 twiddles w^r, r = 1..15, held as SIX values: r = 4a + b, w^r = w^(4a) * w^b with
    // twb = {w, w^2, w^3} and twa = {w^4, w^8, w^12}.  Costs 9 extra complex multiplies per butterfly
    // and frees 18 registers per thread (the overlap-save kernel was two registers short of running
    // without scratch spills, and every spill reload next to a store waits for that store to retire).
    template <int NS, int DIR>
    static BDSP_HD void compute_pre16_split(cpx<T> (&v)[E], const cpx<T>* twa, const cpx<T>* twb)
    {
        static_assert(E == 16, "one radix-16 butterfly per thread");
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            if (r & 3) v[r] = twmul<DIR>(v[r], twb[(r & 3) - 1]);
            if (r >> 2) v[r] = twmul<DIR>(v[r], twa[(r >> 2) - 1]);
        }
        dft<16, DIR>(&v[0]);
    }
    template <int NS, class TW>
    static BDSP_HD void load_twiddles16_split(cpx<T>* twa, cpx<T>* twb, int t, TW tw)
    {
        const int k = t % NS, step = N / (NS * 16);
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            twb[i - 1] = tw(i * k * step);
            twa[i - 1] = tw(4 * i * k * step);
        }
    }

    // The eight twiddle values of dft16_tw for the radix-16 stage with previous product NS (w = w_N^(k N / (16 NS)),
    // k = t mod NS): all of them exact table entries.
    template <int NS, class TW>
    static BDSP_HD void load_twiddles16_fma(cpx<T>* tws, int t, TW tw)
    {
        static_assert(E == 16 && N % 16 == 0, "one radix-16 butterfly per thread");
        const int e = (t % NS) * (N / (NS * 16));
        tws[0] = tw(8 * e);
        tws[1] = tw(4 * e);
        tws[2] = tw(2 * e);
        tws[3] = tw(2 * e + N / 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) tws[4 + j] = tw(e + j * (N / 16));
    }

    // the four held values of expand_twiddles16_fma
    template <int NS, class TW>
    static BDSP_HD void load_twiddles16_fma4(cpx<T>* q, int t, TW tw)
    {
        static_assert(E == 16 && N % 16 == 0, "one radix-16 butterfly per thread");
        const int e = (t % NS) * (N / (NS * 16));
        q[0] = tw(8 * e);
        q[1] = tw(4 * e);
        q[2] = tw(2 * e);
        q[3] = tw(e);
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
    // (EL: the element that crosses -- cpx<T>, or T when a kernel short of LDS exchanges real and imaginary parts one
    // after the other through a buffer of half the size)
    template <int R, int NS, typename EL, typename ARR>
    static BDSP_HD void scatter(const ARR& v, int t, EL* lds)
    {
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            const int j = t + b * NT;
            if (NS % 16 == 0) {
                EL* p = lds + pad((j / NS) * NS * R + (j % NS));
#pragma unroll
                for (int r = 0; r < R; ++r) p[r * (NS / 16) * 17] = v[b * R + r];
            } else if (NS == 1 && R == 16) {
                EL* p = lds + 17 * j;
#pragma unroll
                for (int r = 0; r < R; ++r) p[r] = v[b * R + r];
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) lds[pad(out_index<R, NS>(t, b, r))] = v[b * R + r];
            }
        }
    }

    // Conflict-free layouts for the radix-16 x 16 x 16 chain of 8-byte elements (N = 4096, 256 threads).
    // A wave64 ds_*_b64 access is served in two groups of 32 lanes; with one pad element per 16 a group's
    // 32 consecutive elements span 33 slots = 66 banks and two lanes collide on every gather
    // (*measured*: 20 % of the overlap-save kernel's LDS cycles).  Padding per 32 elements keeps a gather
    // group inside 64 banks, and TWO pad elements per 32 suit both scatters: phys(i) = i + 2*(i >> 5).
    //   exchange A (after the NS = 1 stage): thread j writes 16 contiguous elements, which the hardware
    //     handles as 16-byte pairs -- 16 lanes x 4 dwords must tile the 64 banks, i.e. lane bases that
    //     are even and distinct mod 32 elements: 16 j + 2 (j >> 1) is (one pad per 32 is NOT: measured);
    //   exchange B (after the NS = 16 stage): 16 lanes write a contiguous run, the next 16 lanes a run
    //     272 elements further = 32 banks further.
    // *Measured* (tools/ubench/lds_exchange_rate): 285 -> 244 ns per exchange per CU.
    // Everything stays in base(thread) + constant(register) form.  16-byte elements (f64) keep pad().
    static constexpr bool LAYOUT2 = sizeof(cpx<T>) == 8 && E == 16 && NT == 256 && N == 4096;
    static BDSP_HD void scatter_a(const cpx<T> (&v)[E], int t, cpx<T>* lds)
    {
        if (LAYOUT2) {
            cpx<T>* p = lds + 16 * t + (t & ~1);
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = v[r];
        } else {
            scatter<16, 1>(v, t, lds);
        }
    }
    static BDSP_HD void gather_a(cpx<T> (&v)[E], int t, const cpx<T>* lds)
    {
        if (LAYOUT2) {
            const cpx<T>* p = lds + t + 2 * (t >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = p[r * 272];
        } else {
            gather<16>(v, t, lds);
        }
    }
    static BDSP_HD void scatter_b(const cpx<T> (&v)[E], int t, cpx<T>* lds)
    {
        if (LAYOUT2) {
            cpx<T>* p = lds + 272 * (t >> 4) + (t & 15);
#pragma unroll
            for (int r = 0; r < 16; ++r) p[16 * r + 2 * (r >> 1)] = v[r];
        } else {
            scatter<16, 16>(v, t, lds);
        }
    }
    static BDSP_HD void gather_b(cpx<T> (&v)[E], int t, const cpx<T>* lds)
    {
        if (LAYOUT2) {
            const cpx<T>* p = lds + t + 2 * (t >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = p[r * 272];
        } else {
            gather<16>(v, t, lds);
        }
    }

    // LAYOUT3: exchange A without the two-way WRITE conflict LAYOUT2 still has (PMC: 20 % of the overlap-save
    // kernel's LDS cycles; the analytic bank model of MI355X_MICROARCH.md puts all of them on scatter_a, whose 16-byte
    // store pairs are served in groups of 8 lanes over 32 dword banks: bases 16 j + (j & ~1) put lanes j and j+1 on the
    // same banks).  Row j (the 16 results of thread j) starts at 16 j + 2 (j & 7) + 16 (j >> 3): eight consecutive
    // lanes tile the 32 banks.  A 32-lane READ group then needs its two rows 16 elements apart mod 32, which rows j and
    // j + 8 are: the thread that reads for stage 2 therefore takes column col3(t) (thread bits 4 and 5..7 swapped),
    // and writes exchange B from that column.  Exchange B's layout and gather_b are unchanged, so the transform's
    // input and output register maps stay what they were.  4608 elements instead of 4368.
    static constexpr int LDS_ELEMS3 = 4608;
    static BDSP_HD int col3(int t) { return 16 * ((t >> 5) + 8 * ((t >> 4) & 1)) + (t & 15); }
    static BDSP_HD void scatter_a3(const cpx<T> (&v)[E], int t, cpx<T>* lds)
    {
        cpx<T>* p = lds + 16 * t + 2 * (t & 7) + 16 * (t >> 3);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = v[r];
    }
    static BDSP_HD void gather_a3(cpx<T> (&v)[E], int t, const cpx<T>* lds)
    {
        const cpx<T>* p = lds + 18 * (t >> 5) + 144 * ((t >> 4) & 1) + (t & 15);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = p[r * 288];
    }
    static BDSP_HD void scatter_b3(const cpx<T> (&v)[E], int t, cpx<T>* lds)
    {
        const int c = col3(t);
        cpx<T>* p = lds + 272 * (c >> 4) + (c & 15);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[16 * r + 2 * (r >> 1)] = v[r];
    }

    template <int R, typename EL, typename ARR>
    static BDSP_HD void gather(ARR&& v, int t, const EL* lds)
    {
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            if ((N / R) % 16 == 0) {
                const EL* p = lds + pad(t + b * NT);
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
