// Host-side emulation of the workgroup FFT (basic_dsp_amd/csrc/fft_core.h): threads become a
// loop, barriers become loop boundaries, LDS becomes an array.  Verifies the Stockham index math,
// butterflies and twiddle conventions against a naive O(N^2) DFT without needing a GPU.
#include <array>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../basic_dsp_amd/csrc/fft_core.h"

using namespace bdsp;

template <typename T>
static std::vector<std::complex<double>> naive(const std::vector<cpx<T>>& x, int dir)
{
    size_t n = x.size();
    std::vector<std::complex<double>> out(n);
    for (size_t k = 0; k < n; ++k) {
        std::complex<double> s = 0;
        for (size_t i = 0; i < n; ++i) {
            double a = dir * 2.0 * M_PI * (double)((i * k) % n) / (double)n;
            s += std::complex<double>(x[i].x, x[i].y) * std::complex<double>(cos(a), sin(a));
        }
        out[k] = s;
    }
    return out;
}

// FMA: the radix-16 stages after the first use the twiddled 16-point transform in FMA form (dft16_tw, eight twiddle
// values per thread) instead of fifteen complex multiplies + dft16; PRUNE: the last stage skips outputs X[.. < PRUNE)
template <typename T, int N, int DIR, bool FMA = false, int PRUNE = 0>
static double run()
{
    constexpr int NT = N / 16;
    using F = WgFft<T, N, NT>;
    using P = Radix16Plan<N>;
    std::vector<cpx<T>> x(N), tab(N), out(N);
    srand(N * 7 + DIR);
    for (auto& v : x) v = {(T)(rand() / (double)RAND_MAX * 20 - 10), (T)(rand() / (double)RAND_MAX * 20 - 10)};
    for (int m = 0; m < N; ++m) tab[m] = {(T)cos(-2.0 * M_PI * m / N), (T)sin(-2.0 * M_PI * m / N)};
    auto tw = [&](int m) { return tab[m]; };
    std::vector<cpx<T>> lds(F::LDS_ELEMS);
    std::vector<std::array<cpx<T>, 16>> regs(NT);
    // stage 1: load from "global"
    for (int t = 0; t < NT; ++t) {
        cpx<T>(&v)[16] = *reinterpret_cast<cpx<T>(*)[16]>(regs[t].data());
        for (int r = 0; r < 16; ++r) v[r] = x[F::template in_index<16>(t, 0, r)];
        F::template compute<P::R1, 1, DIR>(v, t, tw);
    }
    if (P::R2 > 1) {
        for (int t = 0; t < NT; ++t) {
            cpx<T>(&v)[16] = *reinterpret_cast<cpx<T>(*)[16]>(regs[t].data());
            F::template scatter<P::R1, 1>(v, t, lds.data());
        }
        for (int t = 0; t < NT; ++t) {
            cpx<T>(&v)[16] = *reinterpret_cast<cpx<T>(*)[16]>(regs[t].data());
            F::template gather<P::R2>(v, t, lds.data());
            if constexpr (FMA && P::R2 == 16) {
                cpx<T> tws[8];
                F::template load_twiddles16_fma<P::R1>(tws, t, tw);
                if constexpr (P::R3 == 1) dft16_tw<DIR, PRUNE>(&v[0], tws); else dft16_tw<DIR>(&v[0], tws);
            } else
            F::template compute<P::R2, P::R1, DIR>(v, t, tw);
        }
    }
    if (P::R3 > 1) {
        for (int t = 0; t < NT; ++t) {
            cpx<T>(&v)[16] = *reinterpret_cast<cpx<T>(*)[16]>(regs[t].data());
            F::template scatter<P::R2, P::R1>(v, t, lds.data());
        }
        for (int t = 0; t < NT; ++t) {
            cpx<T>(&v)[16] = *reinterpret_cast<cpx<T>(*)[16]>(regs[t].data());
            F::template gather<P::R3>(v, t, lds.data());
            if constexpr (FMA && P::R3 == 16) {
                cpx<T> tws[8];
                F::template load_twiddles16_fma<P::R1 * P::R2>(tws, t, tw);
                dft16_tw<DIR, PRUNE>(&v[0], tws);
            } else
            F::template compute<P::R3, P::R1 * P::R2, DIR>(v, t, tw);
        }
    }
    // final: natural order out_index of the last stage
    for (int t = 0; t < NT; ++t) {
        cpx<T>(&v)[16] = *reinterpret_cast<cpx<T>(*)[16]>(regs[t].data());
        if (P::R3 > 1) {
            for (int b = 0; b < 16 / P::R3; ++b)
                for (int r = 0; r < P::R3; ++r)
                    out[F::template out_index<P::R3, P::R1 * P::R2>(t, b, r)] = v[b * P::R3 + r];
        } else if (P::R2 > 1) {
            for (int b = 0; b < 16 / P::R2; ++b)
                for (int r = 0; r < P::R2; ++r)
                    out[F::template out_index<P::R2, P::R1>(t, b, r)] = v[b * P::R2 + r];
        } else {
            for (int r = 0; r < 16; ++r) out[F::template out_index<16, 1>(t, 0, r)] = v[r];
        }
    }
    auto ref = naive<T>(x, DIR);
    double num = 0, den = 0;
    for (int k = 0; k < N; ++k) {
        if (k < PRUNE * (N / 16)) continue; // rows the pruned last stage does not produce
        std::complex<double> d = std::complex<double>(out[k].x, out[k].y) - ref[k];
        num += std::norm(d);
        den += std::norm(ref[k]);
    }
    return sqrt(num / den);
}

template <typename T, int N>
static int check(double tol)
{
    double ef = run<T, N, -1>(), ei = run<T, N, +1>();
    printf("%s N=%5d  fwd rel-L2 %.3e  inv rel-L2 %.3e\n", sizeof(T) == 4 ? "f32" : "f64", N, ef, ei);
    return (ef < tol && ei < tol) ? 0 : 1;
}

template <typename T, int N, int PRUNE>
static int check_fma(double tol)
{
    double ef = run<T, N, -1, true, PRUNE>(), ei = run<T, N, +1, true, PRUNE>();
    printf("%s N=%5d  FMA form, prune %d: fwd rel-L2 %.3e  inv rel-L2 %.3e\n", sizeof(T) == 4 ? "f32" : "f64", N, PRUNE, ef, ei);
    return (ef < tol && ei < tol) ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad += check_fma<float, 256, 0>(5e-7);
    bad += check_fma<float, 4096, 0>(5e-7);
    bad += check_fma<float, 4096, 4>(5e-7);
    bad += check_fma<float, 4096, 8>(5e-7);
    bad += check_fma<float, 4096, 1>(5e-7);
    bad += check_fma<double, 256, 3>(1e-14);
    bad += check_fma<double, 4096, 0>(1e-14);
    bad += check_fma<double, 4096, 5>(1e-14);
    bad += check<float, 16>(5e-7);
    bad += check<float, 32>(5e-7);
    bad += check<float, 64>(5e-7);
    bad += check<float, 128>(5e-7);
    bad += check<float, 256>(5e-7);
    bad += check<float, 512>(5e-7);
    bad += check<float, 1024>(5e-7);
    bad += check<float, 2048>(5e-7);
    bad += check<float, 4096>(5e-7);
    bad += check<double, 16>(1e-14);
    bad += check<double, 256>(1e-14);
    bad += check<double, 2048>(1e-14);
    bad += check<double, 4096>(1e-14);
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
