#!/usr/bin/env python3
"""Randomised differential run of the hot path against the oracle: random lengths (any n, with smooth, prime and
power-of-two ones mixed in), random tap counts, random interpolation factors, both precisions, real and complex.
Seeded; prints the first mismatch and exits non-zero.  usage: fuzz_hot_path.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as orc
from basic_dsp_amd import DspVec, vector as V

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)

def rel(got, ref):
    ref = np.asarray(ref, np.float64); got = np.asarray(got, np.float64)
    d = np.linalg.norm(ref)
    return np.linalg.norm(got - ref) / (d if d > 0 else 1.0)

def pick_n(hi):
    k = rng.integers(0, 5)
    if k == 0: return int(2 ** rng.integers(0, int(np.log2(hi)) + 1))
    if k == 1:  # smooth
        n = 1
        while True:
            f = int(rng.choice([2, 3, 5, 7, 11, 13]))
            if n * f > hi: return max(n, 1)
            n *= f
            if rng.random() < 0.15: return n
    return int(rng.integers(1, hi + 1))

t_end = time.time() + budget
count = 0
while time.time() < t_end:
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    tol = 2e-6 if dtype == np.float32 else 1e-11
    cplx = rng.random() < 0.7
    e = 2 if cplx else 1
    op = rng.integers(0, 4)
    seed = int(rng.integers(1, 1 << 30))
    if op == 0:    # fft / ifft of any length
        n = pick_n(300000)
        x = orc.fill_uniform(2 * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=True)
        which = rng.integers(0, 3)
        if which == 0:
            assert v.plain_fft() == 0
            ref = orc.fft(x.astype(np.float64))
        elif which == 1:
            assert v.fft() == 0
            ref = orc.swap_halves(np.array(orc.fft(x.astype(np.float64))), True, True)
        else:
            f = DspVec(x, is_complex=True, domain=V.FREQ)
            assert f.plain_ifft() == 0
            v = f
            ref = orc.fft(x.astype(np.float64), inverse=True)
        r = rel(v.data(), ref)
        ok, what = r < tol * (4 if n > 4096 else 1), ("fft", which, n, dtype.__name__, r)
    elif op == 1:  # convolve_signal
        n = pick_n(200000)
        m = int(rng.integers(1, min(n, 4200) + 1))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        h = orc.fill_uniform(e * m, seed + 1, -1, 1, dtype) / dtype(m)
        v = DspVec(x, is_complex=cplx)
        assert v.convolve_signal(DspVec(h, is_complex=cplx)) == 0
        first = int(rng.integers(0, max(1, n - 300)))
        cnt = min(300, n - first)
        ref = orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), cplx, first, cnt)
        r = rel(v.data()[e * first:e * (first + cnt)], ref)
        head = rel(v.data()[:e * min(50, n)], orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), cplx, 0, min(50, n)))
        ok, what = max(r, head) < tol * 2, ("conv", n, m, cplx, dtype.__name__, r, head)
    elif op == 2:  # interpolatef
        n = int(rng.integers(30, 60000))
        factor = float(rng.choice([2.0, 3.0, 4.0, 8.0, 1.5, 2.5, 0.75, 5.0]))
        L = int(rng.integers(1, 20))
        fid, ro = (0, 0.0) if rng.random() < 0.5 else (1, 0.35)
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=cplx)
        assert v.interpolatef(fid, factor, 0.0, L, rolloff=ro) == 0
        # the oracle runs in the vector's own precision for the output length and the path choice, in f64 for the values
        # the sampling positions i / factor are computed in T by the reference (and here): for fractional factors the
        # comparison must use the oracle in the SAME precision, an f64 oracle sits 1e-3 away at f32 (position error
        # 6e-8 * 4e4 samples); power-of-two factors have exact positions and are held to the f64 oracle
        exact = factor in (2.0, 4.0, 8.0)   # i / factor is exact in binary floating point only for these
        ref, _path = orc.interpolatef(x.astype(np.float64) if exact else x, cplx, fid, ro, factor, 0.0, L)
        r = rel(v.data(), ref) if len(ref) == len(v.data()) else 1.0
        lim = tol * 3 if exact else (3e-5 if dtype == np.float32 else 1e-10)
        ok, what = r < lim, ("interpolatef", n, factor, L, fid, cplx, dtype.__name__, r)
    else:          # elementwise chain, bit-exact
        n = int(rng.integers(1, 300000))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=cplx)
        a, b = dtype(rng.uniform(-3, 3)), dtype(rng.uniform(-3, 3))
        assert v.scale(float(a)) == 0 and v.offset(float(b)) == 0
        ref = x * a
        ref = ref.astype(dtype)
        if cplx: ref[0::2] = ref[0::2] + b
        else: ref = ref + b
        ok, what = np.array_equal(v.data(), ref.astype(dtype)), ("scale+offset", n, cplx, dtype.__name__)
    count += 1
    if not ok:
        print("MISMATCH", what)
        sys.exit(1)
print("fuzz ok: %d cases in %.0f s" % (count, budget))
