#!/usr/bin/env python3
"""Randomised differential run of the hot path against the oracle: random lengths (any n, with smooth, prime and
power-of-two ones mixed in), random tap counts, random interpolation factors, both precisions, real and complex.
Seeded; prints the first mismatch and exits non-zero.  usage: fuzz_hot_path.py [seconds] [seed] [size scale]
(size scale 8 reaches the 2^21-point two-pass plans and their Bluestein users)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as orc
from basic_dsp_amd import DspVec, DspMat, vector as V
import basic_dsp_amd as bd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1

def rel(got, ref):
    ref = np.asarray(ref, np.float64); got = np.asarray(got, np.float64)
    d = np.linalg.norm(ref)
    return np.linalg.norm(got - ref) / (d if d > 0 else 1.0)

def _reg3_lengths():
    """the lengths of the register-resident mixed-radix kernel (round 6; tools/gen_reg3_table.py, restated)"""
    import itertools
    best = {}
    for t in itertools.combinations_with_replacement([4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25], 3):
        n = t[0] * t[1] * t[2]
        if n > 4096 or n < 300 or n & (n - 1) == 0 or n // min(t) > 512: continue
        best[n] = 1
    R = [4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25]
    for a in R:          # ... and of its two-stage sibling k_mr_reg2: n = R0 R1 < 300
        for b in R:
            if a * b < 300 and (a * b) & (a * b - 1): best[a * b] = 1
    return sorted(best)
REG3 = _reg3_lengths()

def pick_n(hi):
    hi = hi * SCALE
    k = rng.integers(0, 6)
    if k == 0: return int(2 ** rng.integers(0, int(np.log2(hi)) + 1))
    if k == 5:  # a length k_mr_reg3 is built for (every transform family then runs through it, plain and with fused options)
        c = [n for n in REG3 if n <= hi]
        if c: return int(rng.choice(c))
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
ill_conditioned = 0  # raised-cosine cases the reference itself cannot vouch for (judged against exact weights)
while time.time() < t_end:
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    tol = 2e-6 if dtype == np.float32 else 1e-11
    cplx = rng.random() < 0.7
    e = 2 if cplx else 1
    op = rng.integers(0, 14)
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
        # round 5: fractional factors that are not short binary fractions, every roll-off whose second singularity the taps can
        # hit exactly (1 / (2 beta) = 4, 2, 1) next to 0.35, integer and fractional delays -- the fractional-factor kernel's
        # selects and its cancellation-free branch near |2 beta j| = 1
        factor = float(rng.choice([2.0, 3.0, 4.0, 8.0, 1.5, 2.5, 0.75, 5.0, 48.0 / 44.1, 1.25, 44.1 / 48.0, 3.7]))
        L = int(rng.integers(1, 20))
        fid, ro = (0, 0.0) if rng.random() < 0.4 else (1, float(rng.choice([0.35, 0.125, 0.25, 0.5, 0.2])))
        delay = float(rng.choice([0.0, 0.0, 1.0, -2.0, 0.3, 0.5]))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=cplx)
        assert v.interpolatef(fid, factor, delay, L, rolloff=ro) == 0
        # the oracle runs in the vector's own precision for the output length and the path choice, in f64 for the values
        # the sampling positions i / factor are computed in T by the reference (and here): for fractional factors the
        # comparison must use the oracle in the SAME precision, an f64 oracle sits 1e-3 away at f32 (position error
        # 6e-8 * 4e4 samples); power-of-two factors have exact positions and are held to the f64 oracle
        exact = factor in (2.0, 4.0, 8.0) and delay == 0.0   # i / factor is exact in binary floating point only for these
        ref, _path = orc.interpolatef(x.astype(np.float64) if exact else x, cplx, fid, ro, dtype(factor) if not exact else factor, delay, L)
        r = rel(v.data(), ref) if len(ref) == len(v.data()) else 1.0
        lim = tol * 3 if exact else (3e-5 if dtype == np.float32 else 1e-10)
        ok, what = r < lim, ("interpolatef", n, factor, delay, L, fid, ro, cplx, dtype.__name__, r)
        if not ok and fid == 1:
            # a tap next to the raised cosine's second singularity: the reference's expression (and the literal oracle) has
            # no correct digit there; the library's cancellation-free form is held to the oracle's exact-weights mode, and
            # the literal oracle must be the one that is off (oracle/bdsp_oracle.c)
            with orc.exact_weights():
                ref2, _ = orc.interpolatef(x.astype(np.float64) if exact else x, cplx, fid, ro, dtype(factor) if not exact else factor, delay, L)
            r2, rl = rel(v.data(), ref2), rel(ref, ref2)
            ok = r2 < lim and rl > lim / 4
            ill_conditioned += ok
            what = what + ("vs exact weights", r2, "literal oracle vs exact", rl)
    elif op == 4:  # windowed_fft with every window, any length
        n = pick_n(100000)
        if n < 2: n = 2
        w = int(rng.choice([V.WINDOW_TRIANGULAR, V.WINDOW_HAMMING, V.WINDOW_BLACKMAN_HARRIS, V.WINDOW_RECTANGULAR]))
        x = orc.fill_uniform(2 * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=True)
        assert v.windowed_fft(w) == 0
        # the window itself in T, like the reference (Blackman-Harris at its end points is a cancellation of four
        # terms down to 6e-5: an f64 window differs from the f32 one by 1e-3 relative there); the transform in f64
        ref = orc.swap_halves(np.array(orc.fft(orc.apply_window(x, True, w).astype(np.float64))), True, True)
        r = rel(v.data(), ref)
        ok, what = r < tol * 4, ("windowed_fft", w, n, dtype.__name__, r)
    elif op == 5:  # correlate with a padded argument
        n = int(rng.integers(2, 30000))
        x = orc.fill_uniform(2 * n, seed, -10, 10, dtype)
        y = orc.fill_uniform(2 * n, seed + 1, -10, 10, dtype)
        arg = DspVec(y, is_complex=True)
        assert arg.prepare_argument_padded() == 0
        _, ref_arg = orc.prepare_argument(y.astype(np.float64), True)
        v = DspVec(x, is_complex=True)
        code = v.correlate(arg)
        rc, ref = orc.correlate(x.astype(np.float64), ref_arg)
        r = rel(v.data(), ref) if code == rc == 0 else 1.0
        ok, what = r < tol * 4, ("correlate", n, dtype.__name__, code, rc, r)
    elif op == 6:  # interpolate to a random number of points (FFT -> pad / crop -> IFFT), interpolatei
        n = int(rng.integers(8, 20000))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=cplx)
        if rng.random() < 0.5:
            dest = int(rng.integers(max(2, n // 3), 3 * n))
            fid = int(rng.choice([0, 1]))
            code = v.interpolate(fid, dest, 0.0, rolloff=0.35)
            # in the vector's precision: the brick-wall response compares the T-valued axis with 1, and a bin on the
            # edge can fall either way in another precision
            rc, ref = orc.interpolate(x, cplx, fid, 0.35, dest, 0.0)[:2]
            what = ("interpolate", n, dest, fid, cplx, dtype.__name__)
        else:
            fac = int(rng.integers(2, 6))
            fid = int(rng.choice([0, 1]))
            code = v.interpolatei(fid, fac, rolloff=0.35)
            rc, ref = orc.interpolatei(x, cplx, fid, 0.35, fac)[:2]
            what = ("interpolatei", n, fac, fid, cplx, dtype.__name__)
        r = rel(v.data(), ref) if (code == rc == 0 and len(ref) == len(v.data())) else (0.0 if code == rc != 0 else 1.0)
        ok, what = r < (1e-4 if dtype == np.float32 else 1e-9), what + (code, rc, r)
    elif op == 7:  # statistics and the running sum
        n = int(rng.integers(1, 400000))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=cplx)
        st = v.statistics()
        rs = (orc.complex_statistics if cplx else orc.real_statistics)(x.astype(np.float64))
        okst = st["count"] == rs["count"] and st["max_index"] == rs["max_index"] and st["min_index"] == rs["min_index"] \
            and abs(st["sum"] - rs["sum"]) <= (1e-4 if dtype == np.float32 else 1e-9) * (abs(rs["sum"]) + n)
        assert v.cum_sum() == 0
        cs = np.cumsum(x.astype(np.float64).reshape(-1, e), axis=0).reshape(-1)
        okcs = np.max(np.abs(v.data() - cs)) <= (3e-7 if dtype == np.float32 else 1e-13) * (np.max(np.abs(cs)) + 1)
        ok, what = okst and okcs, ("statistics/cum_sum", n, cplx, dtype.__name__, okst, okcs)
    elif op == 8:  # B1 gpu_convolve_vector on host slices, around the threshold of the pipelined-transfer path
        n = int(rng.integers((1 << 20) - 3000, (1 << 20) + 300000)) if rng.random() < 0.7 else int(rng.integers(1, 50000))
        m = int(rng.integers(1, min(n, 3500) + 1))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        h = orc.fill_uniform(e * m, seed + 1, -1, 1, dtype) / dtype(m)
        y, rg = V.gpu_convolve_vector(x, h, cplx)
        v = DspVec(x, is_complex=cplx)
        assert v.convolve_signal(DspVec(h, is_complex=cplx)) == 0
        # the B1 size policy (round 6) declines small jobs whose CPU fallback is the reference's direct form
        thr = bd.lib.bdsp_hip_b1_policy_get(bd._lib.B1_CONV_MIN_WORK_F32 if dtype == np.float32 else bd._lib.B1_CONV_MIN_WORK_F64)
        declined = ((not cplx) or e * m <= 15 or e * n <= 10 * e * m) and n * m < thr
        ok = (y is None) if declined else (y is not None and np.array_equal(y, v.data()))
        what = ("b1 convolve", n, m, cplx, dtype.__name__, "declined expected" if declined else "")
    elif op == 9:  # index moves, bit-exact: any length incl. odd point counts and tiny vectors
        n = int(rng.integers(1, 100000)) if rng.random() < 0.8 else int(rng.integers(1, 12))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        v = DspVec(x, is_complex=cplx)
        k = rng.integers(0, 6)
        if k == 0:
            assert v.swap_halves() == 0; ref = orc.swap_halves(x, cplx, True)
        elif k == 1:
            assert v.reverse() == 0; ref = orc.reverse(x, cplx)
        elif k == 2:
            pts = n + int(rng.integers(1, 2 * n + 3)); opt = int(rng.integers(0, 3))
            code = v.zero_pad(pts, opt); rc, ref = orc.zero_pad(x, cplx, pts, opt, buffered=True)  # the facade calls zero_pad_b
            assert code == rc, (code, rc)
        elif k == 3:
            f = int(rng.integers(2, 6)); assert v.zero_interleave(f) == 0; ref = orc.zero_interleave(x, cplx, f)
        elif k == 4:
            f = int(rng.integers(1, 7)); d = int(rng.integers(0, f)); assert v.decimatei(f, d) == 0; ref = orc.decimatei(x, cplx, f, d)
        else:
            y2 = orc.fill_uniform(e * n, seed + 3, -10, 10, dtype)
            o = int(rng.integers(0, 4)); code = [v.add, v.sub, v.mul, v.div][o](DspVec(y2, is_complex=cplx))
            rc, ref = orc.binary(x, y2, cplx, o); assert code == rc == 0
        got = v.data()
        ok, what = (len(got) == len(ref) and np.array_equal(got, ref)), ("index move / binary", int(k), n, cplx, dtype.__name__)
    elif op == 10:  # complex -> real maps and the math family on random lengths
        n = int(rng.integers(1, 200000))
        x = orc.fill_uniform(2 * n, seed, -3, 3, dtype)
        # (tan / tanh have poles inside the sampled square: near them f32 and f64 evaluations of the same formula part ways)
        name = str(rng.choice(["sqrt", "square", "exp", "sin", "cos", "ln", "sinh", "cosh", "asinh"]))
        v = DspVec(x, is_complex=True)
        assert getattr(v, name)() == 0
        r = rel(v.data(), orc.math(x.astype(np.float64), True, name))
        m2 = DspVec(x, is_complex=True); assert m2.magnitude_squared() == 0
        okm = np.array_equal(m2.data(), orc.complex_to_real(x, 1))
        ok, what = (r < (2e-5 if dtype == np.float32 else 1e-12) and okm), ("math/c2r", name, n, dtype.__name__, r, okm)
    elif op == 11:  # matrix API: every row of a batched operation equals the oracle's result for that row
        rows = int(rng.integers(1, 40)); n = pick_n(20000)
        if n < 2: n = 2
        xs = orc.fill_uniform(2 * n * rows, seed, -10, 10, dtype).reshape(rows, 2 * n)
        mt = DspMat(xs, is_complex=True)
        k = rng.integers(0, 4)
        probe = sorted(set([0, rows - 1, int(rng.integers(0, rows))]))
        m = 0
        if k == 0:
            assert mt.plain_fft() == 0
            refs = {q: orc.fft(xs[q].astype(np.float64)) for q in probe}
        elif k == 1:
            w = int(rng.choice([V.WINDOW_HAMMING, V.WINDOW_TRIANGULAR]))
            assert mt.windowed_fft(w) == 0
            refs = {q: orc.swap_halves(np.array(orc.fft(orc.apply_window(xs[q], True, w).astype(np.float64))), True, True) for q in probe}
        elif k == 2:
            m = int(rng.integers(1, min(n, 1500) + 1))
            h = orc.fill_uniform(2 * m, seed + 9, -1, 1, dtype) / dtype(m)
            assert mt.convolve_signal(DspVec(h, is_complex=True)) == 0
            # the direct form is the yardstick: the reference's own overlap_discard schedule (which the oracle's
            # convolve_signal restates faithfully) leaves a GAP of never-computed outputs for some (N, M), e.g.
            # N = 14744, M = 1186: outputs 7600 .. 10647 (convolution.rs:337, 388-399, 453-458)
            refs = {q: orc.convolve_direct(xs[q].astype(np.float64), h.astype(np.float64), True) for q in probe} if n * m <= 4_000_000 else \
                   {q: orc.overlap_discard(xs[q].astype(np.float64), h.astype(np.float64), orc.next_power_of_two(m), fair=True)[1] for q in probe}
        else:
            assert mt.swap_halves() == 0 and mt.magnitude_squared() == 0
            refs = {q: orc.complex_to_real(orc.swap_halves(xs[q], True, True), 1) for q in probe}
        got = mt.data()
        r = max(rel(got[q], refs[q]) for q in probe)
        ok, what = r < tol * 4, ("matrix", int(k), rows, n, m, dtype.__name__, r, [rel(got[q], refs[q]) for q in probe])
    elif op == 12:  # interpolate_lin / interpolate_hermite: bit-exact against the oracle in the same precision
        n = int(rng.integers(2, 60000))
        x = orc.fill_uniform(n, seed, -10, 10, dtype)
        factor = float(rng.choice([0.5, 1.37, 2.0, 2.5, 3.0, 4.0, 7.0, 0.9]))
        delay = float(rng.choice([0.0, 0.0, 0.25, -0.4]))
        v = DspVec(x)
        herm = rng.random() < 0.5
        assert (v.interpolate_hermite(factor, delay) if herm else v.interpolate_lin(factor, delay)) == 0
        ref = (orc.interpolate_hermite if herm else orc.interpolate_lin)(x, dtype(factor), dtype(delay))
        got = v.data()
        ok, what = (len(got) == len(ref) and np.array_equal(got, ref)), ("interpolate_lin/hermite", bool(herm), n, factor, delay, dtype.__name__)
    elif op == 13:  # convolve(function, ratio, len) with the built-in impulse responses
        n = int(rng.integers(2, 50000))
        x = orc.fill_uniform(e * n, seed, -10, 10, dtype)
        L = int(rng.integers(1, 400)); ratio = float(rng.choice([0.1, 0.25, 0.5, 0.3]))
        fid, ro = (0, 0.0) if rng.random() < 0.5 else (1, 0.35)
        v = DspVec(x, is_complex=cplx)
        assert v.convolve(fid, ratio, L, rolloff=ro) == 0
        ref = orc.convolve_function(x.astype(np.float64), cplx, fid, ro, ratio, L)
        r = rel(v.data(), ref)
        ok, what = r < tol * 2, ("convolve(function)", n, L, ratio, fid, cplx, dtype.__name__, r)
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
print("fuzz ok: %d cases in %.0f s (seed %s, size scale %d; %d raised-cosine cases next to the second singularity judged against exact weights)" %
      (count, budget, sys.argv[2] if len(sys.argv) > 2 else "12345", SCALE, ill_conditioned))
