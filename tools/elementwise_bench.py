#!/usr/bin/env python3
"""Throughput of the elementwise / window / index-move operations on a 16M-point complex f32 vector."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as orc
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec, vector as V
n = 1 << 24
x = orc.fill_uniform(2 * n, 1, 1, 10, np.float32)
other = DspVec(x, is_complex=True)
def bench(name, make, op, bytes_moved, reps=10):
    """Ops that change the length run on the same handle every time: set_len() restores the original
    length and metadata-free state without touching the (already grown) buffers."""
    v = make()
    n0, c0 = len(v), v.is_complex()
    code = op(v)
    assert code == 0, (name, code)
    bd.lib.bdsp_hip_synchronize(None)
    ts = []
    for k in range(reps):
        if len(v) != n0 or v.is_complex() != c0:
            v = make() if v.is_complex() != c0 else v
            if len(v) != n0:
                v._fn("set_len")(v._h, n0)
            if v.is_complex() != c0:
                op(v); v = make(); op(v)  # complex->real ops: time a fresh handle whose buffers exist
                continue
        bd.lib.bdsp_hip_synchronize(None)
        t0 = time.perf_counter()
        op(v)
        bd.lib.bdsp_hip_synchronize(None)
        ts.append(time.perf_counter() - t0)
    if not ts:
        # number-space changing ops: buffers of a fresh handle are large enough (same or smaller output)
        for k in range(reps):
            v = make()
            bd.lib.bdsp_hip_synchronize(None)
            t0 = time.perf_counter()
            op(v)
            bd.lib.bdsp_hip_synchronize(None)
            ts.append(time.perf_counter() - t0)
    us = sorted(ts)[len(ts) // 2] * 1e6
    print("%-28s %8.1f us  %6.0f GB/s" % (name, us, bytes_moved / us / 1e3))
cv = lambda: DspVec(x, is_complex=True)
rv = lambda: DspVec(x[: n], is_complex=False)
B = 8 * n
bench("scale (real factor)", cv, lambda v: v.scale(1.0001), 2 * B)
bench("scale (complex factor)", cv, lambda v: v.scale(1.0 + 0.001j), 2 * B)
bench("offset", cv, lambda v: v.offset(0.5), 2 * B)
bench("conj", cv, lambda v: v.conj(), 2 * B)
bench("mul (vector)", cv, lambda v: v.mul(other), 3 * B)
bench("add (vector)", cv, lambda v: v.add(other), 3 * B)
bench("multiply_complex_exponential", cv, lambda v: v.multiply_complex_exponential(0.001, 0.5), 2 * B)
bench("apply_window(Hamming)", cv, lambda v: v.apply_window(V.WINDOW_HAMMING), 2 * B)
bench("magnitude", cv, lambda v: v.magnitude(), B + B // 2)
bench("phase", cv, lambda v: v.phase(), B + B // 2)
bench("to_real", cv, lambda v: v.to_real(), B + B // 2)
bench("swap_halves", cv, lambda v: v.swap_halves(), 2 * B)
bench("reverse", cv, lambda v: v.reverse(), 2 * B)
bench("zero_pad (x2, surround)", cv, lambda v: v.zero_pad(2 * n, V.PAD_SURROUND), 3 * B)
bench("zero_interleave (x2)", cv, lambda v: v.zero_interleave(2), 3 * B)
bench("to_complex (real 16M)", rv, lambda v: v.to_complex(), B // 2 + B)
bench("decimatei (/2)", cv, lambda v: v.decimatei(2, 0), B + B // 2)
# reductions: read-only passes (the vector is not modified, the handle is reused)
def bench_ro(name, v, op, bytes_moved, reps=10):
    op(v)
    bd.lib.bdsp_hip_synchronize(None)
    ts = []
    for k in range(reps):
        t0 = time.perf_counter()
        op(v)
        bd.lib.bdsp_hip_synchronize(None)
        ts.append(time.perf_counter() - t0)
    us = sorted(ts)[len(ts) // 2] * 1e6
    print("%-28s %8.1f us  %6.0f GB/s" % (name, us, bytes_moved / us / 1e3))
c, r = cv(), rv()
bench_ro("statistics (complex 16M)", c, lambda v: v.statistics(), B)
bench_ro("statistics (real 16M)", r, lambda v: v.statistics(), B // 2)
bench_ro("sum_sq (complex 16M)", c, lambda v: v.sum_sq(), B)
bench_ro("dot_product (complex 16M)", c, lambda v: v.dot_product(other), 2 * B)
bench_ro("statistics_split(4)", c, lambda v: v.statistics_split(4), B)
# per-element math family, differences / running sums, pairs, split / merge
bench("sqrt (complex)", cv, lambda v: v.sqrt(), 2 * B)
bench("square (complex)", cv, lambda v: v.square(), 2 * B)
bench("exp (complex)", cv, lambda v: v.exp(), 2 * B)
bench("sin (complex)", cv, lambda v: v.sin(), 2 * B)
bench("ln (complex)", cv, lambda v: v.ln(), 2 * B)
bench("powf(2.5) (complex)", cv, lambda v: v.powf(2.5), 2 * B)
bench("atanh (complex)", cv, lambda v: v.atanh(), 2 * B)
bench("sin (real 16M)", rv, lambda v: v.sin(), B)
bench("abs (real 16M)", rv, lambda v: v.abs(), B)
bench("cum_sum (complex)", cv, lambda v: v.cum_sum(), 3 * B)
bench("cum_sum (real 16M)", rv, lambda v: v.cum_sum(), 3 * B // 2)
bench("diff_with_start (complex)", cv, lambda v: v.diff_with_start(), 2 * B)
parts = [DspVec(dtype=np.float32, length=0, is_complex=True) for _ in range(4)]
bench_ro("split_into(4)", c, lambda v: v.split_into(parts), 2 * B)
m = DspVec(dtype=np.float32, length=0, is_complex=True)
bench_ro("merge(4)", m, lambda v: v.merge(parts), 2 * B)
ga, gb = DspVec(dtype=np.float32, length=0), DspVec(dtype=np.float32, length=0)
bench_ro("get_mag_phase", c, lambda v: v.get_mag_phase(ga, gb), 2 * B)   # includes the clone the wrapper makes
bench_ro("set_mag_phase", m, lambda v: v.set_mag_phase(ga, gb), 2 * B)
u = DspVec(x[: 1 << 20])
bench_ro("unwrap (real 1M, serial)", u, lambda v: v.unwrap(7.0), (1 << 20) * 8, reps=3)
