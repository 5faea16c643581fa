// mr_dft.h -- radix-R butterflies in registers for the mixed-radix transforms (mixed_radix.hip, mixed_radix_reg3*.hip):
// base radices 2/3/4/5/7, odd radices 11/13 in symmetric form, 8/16 from fft_core.h, and composite radices
// 6/9/10/12/14/15/20/25 = RA x RB butterflied in registers (constants w_R^(m2 k1) at compile time).
// DIR = -1 forward, +1 inverse; natural order in, natural order out.
#pragma once
#include "bdsp_internal.h"

namespace bdsp {

// ---- small DFTs, natural order; DIR = -1 forward, +1 inverse ---------------------------------------------------
template <int DIR, typename C> __device__ __forceinline__ void mr_dft3(C* v)
{
    using T = typename real_of<C>::type;
    const T h = (T)0.86602540378443864676; // sin(pi/3)
    C t1 = cadd(v[1], v[2]);
    C m1 = csub(v[0], cscale(t1, (T)0.5));
    C m2 = mul_dir_i<DIR>(cscale(csub(v[1], v[2]), h)); // (-/+ i) sin60 (v1 - v2)
    v[0] = cadd(v[0], t1);
    v[1] = cadd(m1, m2);
    v[2] = csub(m1, m2);
}
template <int DIR, typename C> __device__ __forceinline__ void mr_dft5(C* v)
{
    using T = typename real_of<C>::type;
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410; // cos(2pi/5), cos(4pi/5)
    const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;  // sin(2pi/5), sin(4pi/5)
    C t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    C a1 = cadd(v[0], cadd(cscale(t1, c1), cscale(t2, c2)));
    C a2 = cadd(v[0], cadd(cscale(t1, c2), cscale(t2, c1)));
    C b1 = mul_dir_i<DIR>(cadd(cscale(t3, s1), cscale(t4, s2)));
    C b2 = mul_dir_i<DIR>(csub(cscale(t3, s2), cscale(t4, s1)));
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(a1, b1); v[4] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[3] = csub(a2, b2);
}
template <int DIR, typename C> __device__ __forceinline__ void mr_dft7(C* v)
{
    using T = typename real_of<C>::type;
    const T c1 = (T)0.62348980185873353053, c2 = (T)-0.22252093395631440429, c3 = (T)-0.90096886790241912624;
    const T s1 = (T)0.78183148246802980871, s2 = (T)0.97492791218182360702, s3 = (T)0.43388373911755812048;
    C t1 = cadd(v[1], v[6]), t2 = cadd(v[2], v[5]), t3 = cadd(v[3], v[4]);
    C u1 = csub(v[1], v[6]), u2 = csub(v[2], v[5]), u3 = csub(v[3], v[4]);
    // a_k = v0 + sum_m cos(2 pi k m / 7) t_m,   b_k = sum_m sin(2 pi k m / 7) u_m
    C a1 = cadd(v[0], cadd(cscale(t1, c1), cadd(cscale(t2, c2), cscale(t3, c3))));
    C a2 = cadd(v[0], cadd(cscale(t1, c2), cadd(cscale(t2, c3), cscale(t3, c1))));
    C a3 = cadd(v[0], cadd(cscale(t1, c3), cadd(cscale(t2, c1), cscale(t3, c2))));
    C b1 = mul_dir_i<DIR>(cadd(cscale(u1, s1), cadd(cscale(u2, s2), cscale(u3, s3))));
    C b2 = mul_dir_i<DIR>(cadd(cscale(u1, s2), csub(cscale(u2, -s3), cscale(u3, s1))));
    C b3 = mul_dir_i<DIR>(cadd(cscale(u1, s3), cadd(cscale(u2, -s1), cscale(u3, s2))));
    v[0] = cadd(v[0], cadd(t1, cadd(t2, t3)));
    v[1] = cadd(a1, b1); v[6] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[5] = csub(a2, b2);
    v[3] = cadd(a3, b3); v[4] = csub(a3, b3);
}
// odd radix R = 2h+1 (11, 13) in the same symmetric form, coefficients looked up by (k m) mod R at compile time
template <int R> struct MrTrig;
template <> struct MrTrig<11> {
    static constexpr double c[11] = {1, 0.84125353283118120551, 0.41541501300188643508, -0.1423148382732850048, -0.65486073394528498959, -0.95949297361449736865, -0.95949297361449747967, -0.65486073394528521163, -0.14231483827328522684, 0.41541501300188604651, 0.84125353283118120551};
    static constexpr double s[11] = {0, 0.54064081745559755543, 0.90963199535451833011, 0.98982144188093279524, 0.7557495743542582689, 0.28173255684142967104, -0.28173255684142939348, -0.75574957435425815788, -0.98982144188093268422, -0.90963199535451855215, -0.54064081745559744441};
};
template <> struct MrTrig<13> {
    static constexpr double c[13] = {1, 0.88545602565320991051, 0.56806474673115592289, 0.12053668025532300601, -0.35460488704253545489, -0.74851074817110119231, -0.9709418174260520118, -0.97094181742605212282, -0.74851074817110130333, -0.35460488704253589898, 0.12053668025532320029, 0.56806474673115481266, 0.88545602565321002153};
    static constexpr double s[13] = {0, 0.46472317204376850652, 0.82298386589365635224, 0.99270887409805397272, 0.93501624268541483342, 0.66312265824079519305, 0.23931566428755768339, -0.23931566428755743359, -0.66312265824079497101, -0.9350162426854147224, -0.99270887409805397272, -0.82298386589365701838, -0.4647231720437683955};
};
template <> struct MrTrig<6> {
    static constexpr double c[6] = {1, 0.50000000000000011102, -0.49999999999999977796, -1, -0.50000000000000044409, 0.50000000000000011102};
    static constexpr double s[6] = {0, 0.86602540378443859659, 0.86602540378443870761, 1.2246467991473532072e-16, -0.86602540378443837454, -0.86602540378443859659};
};
template <> struct MrTrig<9> {
    static constexpr double c[9] = {1, 0.76604444311897801345, 0.17364817766693041445, -0.49999999999999977796, -0.93969262078590831688, -0.93969262078590842791, -0.50000000000000044409, 0.17364817766692997036, 0.76604444311897779141};
    static constexpr double s[9] = {0, 0.6427876096865392519, 0.98480775301220802032, 0.86602540378443870761, 0.34202014332566887944, -0.3420201433256686574, -0.86602540378443837454, -0.98480775301220813134, -0.64278760968653958496};
};
template <> struct MrTrig<10> {
    static constexpr double c[10] = {1, 0.80901699437494745126, 0.30901699437494745126, -0.30901699437494734024, -0.80901699437494734024, -1, -0.80901699437494756229, -0.30901699437494756229, 0.30901699437494722922, 0.80901699437494734024};
    static constexpr double s[10] = {0, 0.5877852522924731371, 0.95105651629515353118, 0.9510565162951536422, 0.58778525229247324813, 1.2246467991473532072e-16, -0.58778525229247302608, -0.95105651629515353118, -0.9510565162951536422, -0.58778525229247335915};
};
template <> struct MrTrig<12> {
    static constexpr double c[12] = {1, 0.86602540378443870761, 0.50000000000000011102, 6.1232339957367660359e-17, -0.49999999999999977796, -0.86602540378443870761, -1, -0.86602540378443881863, -0.50000000000000044409, -1.8369701987210296875e-16, 0.50000000000000011102, 0.86602540378443837454};
    static constexpr double s[12] = {0, 0.49999999999999994449, 0.86602540378443859659, 1, 0.86602540378443870761, 0.49999999999999994449, 1.2246467991473532072e-16, -0.49999999999999972244, -0.86602540378443837454, -1, -0.86602540378443859659, -0.50000000000000044409};
};
template <> struct MrTrig<14> {
    static constexpr double c[14] = {1, 0.900968867902419146, 0.62348980185873359439, 0.22252093395631444839, -0.22252093395631433737, -0.62348980185873348336, -0.90096886790241903498, -1, -0.900968867902419146, -0.62348980185873370541, -0.22252093395631458717, 0.22252093395631333816, 0.62348980185873337234, 0.90096886790241936804};
    static constexpr double s[14] = {0, 0.4338837391175581204, 0.78183148246802980363, 0.97492791218182361934, 0.97492791218182361934, 0.78183148246802991466, 0.43388373911755823142, 1.2246467991473532072e-16, -0.43388373911755800938, -0.78183148246802969261, -0.97492791218182361934, -0.97492791218182384139, -0.78183148246802991466, -0.43388373911755750978};
};
template <> struct MrTrig<15> {
    static constexpr double c[15] = {1, 0.9135454576426008666, 0.66913060635885823757, 0.30901699437494745126, -0.10452846326765333207, -0.49999999999999977796, -0.80901699437494734024, -0.97814760073380568883, -0.97814760073380568883, -0.80901699437494756229, -0.50000000000000044409, -0.10452846326765423413, 0.30901699437494722922, 0.66913060635885845961, 0.91354545764260097762};
    static constexpr double s[15] = {0, 0.40673664307580015276, 0.7431448254773941331, 0.95105651629515353118, 0.99452189536827340088, 0.86602540378443870761, 0.58778525229247324813, 0.20791169081775931482, -0.20791169081775906502, -0.58778525229247302608, -0.86602540378443837454, -0.99452189536827328986, -0.9510565162951536422, -0.74314482547739402207, -0.40673664307580015276};
};
template <> struct MrTrig<20> {
    static constexpr double c[20] = {1.0, 0.95105651629515357212, 0.8090169943749474241, 0.58778525229247312917, 0.3090169943749474241, 0.0, -0.3090169943749474241, -0.58778525229247312917, -0.8090169943749474241, -0.95105651629515357212, -1.0, -0.95105651629515357212, -0.8090169943749474241, -0.58778525229247312917, -0.3090169943749474241, 0.0, 0.3090169943749474241, 0.58778525229247312917, 0.8090169943749474241, 0.95105651629515357212};
    static constexpr double s[20] = {0.0, 0.3090169943749474241, 0.58778525229247312917, 0.8090169943749474241, 0.95105651629515357212, 1.0, 0.95105651629515357212, 0.8090169943749474241, 0.58778525229247312917, 0.3090169943749474241, 0.0, -0.3090169943749474241, -0.58778525229247312917, -0.8090169943749474241, -0.95105651629515357212, -1.0, -0.95105651629515357212, -0.8090169943749474241, -0.58778525229247312917, -0.3090169943749474241};
};
template <> struct MrTrig<25> {
    static constexpr double c[25] = {1.0, 0.96858316112863111949, 0.87630668004386358731, 0.72896862742141152315, 0.53582679497899661827, 0.3090169943749474241, 0.062790519529313376076, -0.18738131458572463054, -0.42577929156507264886, -0.63742398974868971018, -0.8090169943749474241, -0.92977648588825140366, -0.99211470131447783105, -0.99211470131447783105, -0.92977648588825140366, -0.8090169943749474241, -0.63742398974868971018, -0.42577929156507264886, -0.18738131458572463054, 0.062790519529313376076, 0.3090169943749474241, 0.53582679497899661827, 0.72896862742141152315, 0.87630668004386358731, 0.96858316112863111949};
    static constexpr double s[25] = {0.0, 0.24868988716485478824, 0.48175367410171527499, 0.68454710592868867373, 0.84432792550201507855, 0.95105651629515357212, 0.99802672842827156195, 0.98228725072868868109, 0.90482705246601952771, 0.7705132427757892308, 0.58778525229247312917, 0.36812455268467795916, 0.12533323356430424537, -0.12533323356430424537, -0.36812455268467795916, -0.58778525229247312917, -0.7705132427757892308, -0.90482705246601952771, -0.98228725072868868109, -0.99802672842827156195, -0.95105651629515357212, -0.84432792550201507855, -0.68454710592868867373, -0.48175367410171527499, -0.24868988716485478824};
};
template <int R, int DIR, typename C> __device__ __forceinline__ void mr_dft_odd(C* v)
{
    using T = typename real_of<C>::type;
    constexpr int H = (R - 1) / 2;
    C t[H], u[H], x0 = v[0];
#pragma unroll
    for (int m = 1; m <= H; ++m) { t[m - 1] = cadd(v[m], v[R - m]); u[m - 1] = csub(v[m], v[R - m]); x0 = cadd(x0, t[m - 1]); }
    C a[H], b[H];
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        C ak = v[0], bk = C{(T)0, (T)0};
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            ak = cadd(ak, cscale(t[m - 1], (T)MrTrig<R>::c[(k * m) % R]));
            bk = cadd(bk, cscale(u[m - 1], (T)MrTrig<R>::s[(k * m) % R]));
        }
        a[k - 1] = ak;
        b[k - 1] = mul_dir_i<DIR>(bk);
    }
    v[0] = x0;
#pragma unroll
    for (int k = 1; k <= H; ++k) { v[k] = cadd(a[k - 1], b[k - 1]); v[R - k] = csub(a[k - 1], b[k - 1]); }
}
template <int R, int DIR, typename C> __device__ __forceinline__ void mr_dft_base(C* v)
{
    if constexpr (R == 2) dft2<DIR>(v[0], v[1]);
    else if constexpr (R == 3) mr_dft3<DIR>(v);
    else if constexpr (R == 4) dft4<DIR>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 5) mr_dft5<DIR>(v);
    else mr_dft7<DIR>(v);
}
// composite radix R = RA * RB in registers (6, 9, 10, 12, 14, 15): m = RB m1 + m2, k = k1 + RA k2,
//   DFT_RA over m1 for every m2, constants w_R^(m2 k1), DFT_RB over m2 for every k1 -- fewer LDS round trips and
//   barriers than separate stages (1000 = 10 10 10 instead of 4 2 5 5 5)
template <int RA, int RB, int DIR, typename C> __device__ __forceinline__ void mr_dft_comp(C* v)
{
    using T = typename real_of<C>::type;
    constexpr int R = RA * RB;
    C a[R];
#pragma unroll
    for (int m2 = 0; m2 < RB; ++m2) {
        C u[RA];
#pragma unroll
        for (int m1 = 0; m1 < RA; ++m1) u[m1] = v[RB * m1 + m2];
        mr_dft_base<RA, DIR>(u);
#pragma unroll
        for (int k1 = 0; k1 < RA; ++k1) {
            const int j = (m2 * k1) % R;
            a[k1 * RB + m2] = j == 0 ? u[k1] : twmul<DIR>(u[k1], C{(T)MrTrig<R>::c[j], (T)-MrTrig<R>::s[j]});
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < RA; ++k1) {
        C u[RB];
#pragma unroll
        for (int m2 = 0; m2 < RB; ++m2) u[m2] = a[k1 * RB + m2];
        mr_dft_base<RB, DIR>(u);
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2) v[k1 + RA * k2] = u[k2];
    }
}
template <int R, int DIR, typename C> __device__ __forceinline__ void mr_dft(C* v)
{
    if constexpr (R == 2 || R == 3 || R == 4 || R == 5 || R == 7) mr_dft_base<R, DIR>(v);
    else if constexpr (R == 8) dft8<DIR>(v);
    else if constexpr (R == 16) dft16<DIR>(v);
    else if constexpr (R == 6) mr_dft_comp<2, 3, DIR>(v);
    else if constexpr (R == 9) mr_dft_comp<3, 3, DIR>(v);
    else if constexpr (R == 10) mr_dft_comp<2, 5, DIR>(v);
    else if constexpr (R == 12) mr_dft_comp<4, 3, DIR>(v);
    else if constexpr (R == 14) mr_dft_comp<2, 7, DIR>(v);
    else if constexpr (R == 15) mr_dft_comp<3, 5, DIR>(v);
    else if constexpr (R == 20) mr_dft_comp<4, 5, DIR>(v); // (20 and 25: the register-resident kernel k_mr_reg3 only)
    else if constexpr (R == 25) mr_dft_comp<5, 5, DIR>(v);
    else mr_dft_odd<R, DIR>(v);
}

} // namespace bdsp
