#!/usr/bin/env python3
"""Times every BASELINE.json config on one MI355X through the B3 device API (data resident in HBM)
and prints one JSON line per config with its algorithmic-bytes roofline fraction.
Run on the GPU box: python tools/bench_configs.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
from basic_dsp_amd._lib import FFT_SHIFT_OUT, FFT_MAGNITUDE
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg() if "--cpu-only" not in sys.argv else None
flag = C.c_int(0)
PEAK = 8000.0

QUICK = "--quick" in sys.argv  # a few launches per config only: for the rocprofv3 --pmc passes (tools/profile_round.sh)
NO_CPU = QUICK or "--no-cpu" in sys.argv  # the CPU column (round 6) costs ~1 minute of host time: not under the profiler


def cpu_rows():
    """The CPU baseline beside every BASELINE config, on THIS box's host cores, in this run:
      reference_1core  the oracle (kind "port": the C restatement of the reference algorithm, tests/oracle_lib.py) on one
                       thread, in the reference's own schedule -- separate passes, plan per call; one thread is the
                       reference's default MultiCoreSettings (multicore_support/threading.rs:210-217)
      fair_allcores    the same work with the FFT butterflies / convolution blocks spread over all granted cores (OpenMP in
                       the oracle; `parallel()` would use half of them, threading.rs:220-231); null where the oracle has no
                       threaded form of the operation
      numpy_sanity     numpy / scipy (pocketfft) doing the same job: NOT the reference and not the port -- it shows that the
                       port's radix-2 loop is not a straw man
    Every leg runs three times: `us` is the median, `us_min` the fastest.  Units per second use the config's own unit."""
    import time as _t
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib as orc
    import bench
    import scipy.signal as ss
    cores = bench.usable_cores()

    def t3(fn, reps=3):
        v = []
        for _ in range(reps):
            t0 = _t.perf_counter(); fn(); v.append((_t.perf_counter() - t0) * 1e6)
        v.sort()
        return {"us": round(v[len(v) // 2], 1), "us_min": round(v[0], 1)}

    def obj(units, unit_name, ref, allc, npy, sample, scale=1.0):
        """scale: the sample is 1/scale of the config (a prefix): microseconds are scaled up to the whole config"""
        def leg(d):
            if d is None:
                return None
            us = d["us"] * scale
            return {"us": round(us, 1), "us_min": round(d["us_min"] * scale, 1), "M%s_per_s" % unit_name: round(units / us, 3)}
        return {"kind": "port", "cores": cores, "reference_1core": leg(ref), "fair_allcores": leg(allc),
                "numpy_sanity": dict(leg(npy), note="numpy / scipy (pocketfft), NOT the reference") if npy else None,
                "sample": sample, "reps": 3, "statistic": "median (us) and fastest (us_min) of three runs"}
    rows = {}
    # C1: scale() then offset(), two passes over 65 536 real f32 (general/elementary.rs:283-640)
    x = orc.fill_uniform(65536, 201601171, -10, 10, np.float32)
    rows["C1 "] = obj(65536, "samples", t3(lambda: orc.real_offset(orc.real_scale(x, 2.5), -1.25), 5), None,
                      t3(lambda: x * np.float32(2.5) + np.float32(-1.25), 5), "the whole config")
    # C2: fft -> magnitude, 1M complex f32, two passes (time_freq/time_to_freq.rs:126-144, complex_to_real.rs:365-478)
    n = 1 << 20
    x = orc.fill_uniform(2 * n, 201601172, -10, 10, np.float32)
    xc = x.view(np.complex64)
    rows["C2 "] = obj(n, "points", t3(lambda: orc.magnitude(orc.fft(x))), t3(lambda: orc.magnitude(orc.fft_pow2_mt(x, False, cores))),
                      t3(lambda: np.abs(np.fft.fft(xc))), "the whole config")
    # C3 / FFT16M: bench.py's CPU baseline on a 2^22-point prefix of the 16M-point vector, split into its two halves
    n, m, sp_ = 1 << 24, 1024, 1 << 22
    b = bench.cpu_baseline(n, m, sp_)
    def mk(b_, *legs):
        return {"us": sum(b_["seconds"][l] for l in legs) * 1e6, "us_min": sum(min(b_["seconds_runs"][l]) for l in legs) * 1e6}
    rows["C3 complex f32 16M"] = obj(n, "samples", mk(b, "reference_overlap_discard_1core"), mk(b, "fair_overlap_save_allcores"), mk(b, "numpy_oaconvolve"),
                                     "%d-point prefix, scaled to 16M points; reference_1core = the reference's overlap_discard schedule incl. its scalar tail, "
                                     "fair_allcores = tail-free overlap-save on all cores" % sp_, n / sp_)
    rows["C3 as ONE"] = rows["C3 complex f32 16M"]
    rows["FFT complex f32 16M"] = obj(n, "points", mk(b, "fft_1core"), mk(b, "fft_allcores"), mk(b, "numpy_fft"),
                                      "%d-point transform, scaled to 16M points by N log N" % sp_, n / sp_ * 24.0 / 22.0)
    # C4a: apply_window(Hann) -> fft -> swap_halves, three separate passes like time_to_freq.rs:167-175; 4M complex f64
    n = 1 << 22
    x = orc.fill_uniform(2 * n, 201601174, -10, 10, np.float64)
    xc = x.view(np.complex128)
    rows["C4a "] = obj(n, "points", t3(lambda: orc.swap_halves(orc.fft(orc.apply_window(x, True, 4, 0.5)), True)),
                       t3(lambda: orc.swap_halves(orc.fft_pow2_mt(orc.apply_window(x, True, 4, 0.5), False, cores), True)),
                       t3(lambda: np.fft.fftshift(np.fft.fft(xc * np.hanning(n)))), "the whole config")
    # C4b: interpolatef(RC 0.35, x4, conv_len 12) on a 2^18-point prefix, scaled (time_freq/interpolation.rs:387-482)
    pre = 1 << 18
    xp = x[:2 * pre].copy()
    taps_rc = np.array([orc.conv_time(1, 0.35, j / 4.0, np.float64) for j in range(-12 * 4, 12 * 4 + 1)])
    rows["C4b "] = obj(n, "input_points", t3(lambda: orc.interpolatef(xp, True, 1, 0.35, 4.0, 0.0, 12)), None,
                       t3(lambda: ss.upfirdn(taps_rc, xp.view(np.complex128), up=4)),
                       "%d-point prefix, scaled to 4M points" % pre, n / pre)
    # C5: 8 whole vectors of the 64 per GPU, one after the other like the matrix crate's row loop, scaled to 64
    b = bench.cpu_baseline(1 << 20, 1024, 8 << 20, vectors=64)
    rows["C5/GPU"] = obj(64 << 20, "samples", mk(b, "reference_overlap_discard_1core", "fft_1core"), mk(b, "fair_overlap_save_allcores", "fft_allcores"),
                         mk(b, "numpy_oaconvolve", "numpy_fft"), "8 whole vectors of the 64, scaled; convolve_signal -> fft each", 8.0)
    return rows


CPU = {} if NO_CPU else cpu_rows()
if "--cpu-only" in sys.argv:  # (runs without a GPU)
    for k_, v_ in CPU.items():
        print(json.dumps({"config": k_.strip(), "cpu_baseline": v_}))
    sys.exit(0)

def timeit(fn, iters=20):
    # untimed pre-warm: the clock needs tens of milliseconds of load to settle (see bench.py)
    import time as _t
    if QUICK: iters = 2
    t0 = _t.perf_counter(); k = 0
    while _t.perf_counter() - t0 < (0.0 if QUICK else 0.15):
        for _ in range(10): fn(k); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): fn(i)
    lib.bdsp_hip_event_record(e1, sp)
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3

_EV_OVERHEAD = None


def event_overhead_us():
    """what an event pair costs by itself on this stream (the smallest of twenty samples)"""
    global _EV_OVERHEAD
    if _EV_OVERHEAD is None:
        v = []
        ms = C.c_float(0)
        for _ in range(20):
            a, b = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
            lib.bdsp_hip_event_record(a, sp); lib.bdsp_hip_event_record(b, sp)
            lib.bdsp_hip_event_elapsed_ms(a, b, C.byref(ms)); v.append(ms.value * 1e3)
            lib.bdsp_hip_event_destroy(a); lib.bdsp_hip_event_destroy(b)
        _EV_OVERHEAD = min(v)
    return _EV_OVERHEAD


def timeit_fresh(call, pristine, iters=20, scratch=None):
    """For calls that CLOBBER their input (the in-place transforms): `call(buf, scr)` runs on a buffer that holds valid data and
    is used exactly ONCE in the timed loop -- `iters` copies of `pristine` are made beforehand (288 GB of HBM: room is not
    the problem), so every call reads a cold, valid input and nothing untimed runs between the calls; the pre-warm runs on
    three more buffers that are restored before each use.  One event pair around the loop.
    Round 5: every call also gets a SCRATCH buffer of its own (`scratch` = a tensor to model them on), as cold as its input
    -- a B2 handle's trade buffer is.  With one shared scratch a transform that leaves its result in the scratch buffer
    never writes a result back to HBM in such a loop (the next call overwrites it in the Infinity Cache), one that leaves
    it in the rotating input does: the loop then ranks plans by where they put the result, not by what they cost.
    Round 4: a loop of in-place transforms on the same buffers feeds each call the previous call's output -- the values
    grow by sqrt(n) per call and are inf / NaN long before the timed region, and kernels run measurably faster on such
    constant bit patterns than on data (config C4b, whose inputs config C4a had left that way: 54-56 us against 68-74)."""
    import time as _t
    if QUICK: iters = 2
    warm = [pristine.clone() for _ in range(3)]
    bufs = [pristine.clone() for _ in range(iters)]
    scrs = [torch.zeros_like(scratch) for _ in range(iters)] if scratch is not None else [None] * iters
    t0 = _t.perf_counter(); k = 0
    while _t.perf_counter() - t0 < (0.0 if QUICK else 0.15):
        for _ in range(5):
            warm[k % 3].copy_(pristine); call(warm[k % 3], scratch); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): call(bufs[i], scrs[i] if scrs[i] is not None else scratch)
    lib.bdsp_hip_event_record(e1, sp)
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    del bufs, warm, scrs
    return ms.value / iters * 1e3


def timeit_hot(call, pristine, iters=20, scratch=None):
    """The same call on an input that was written just before it (an untimed copy from `pristine` into one of three buffers
    right before every call: the input is then in the caches, as after a producer kernel); one event pair per call, median
    of the deltas minus the cost of an empty pair."""
    if QUICK: return None
    bufs = [pristine.clone() for _ in range(3)]
    for k in range(20):
        bufs[k % 3].copy_(pristine); call(bufs[k % 3], scratch)
    pairs = []
    for i in range(iters):
        bufs[i % 3].copy_(pristine)
        a, b = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(a, sp); call(bufs[i % 3], scratch); lib.bdsp_hip_event_record(b, sp)
        pairs.append((a, b))
    torch.cuda.synchronize()
    ms = C.c_float(0); d = []
    for a, b in pairs:
        lib.bdsp_hip_event_elapsed_ms(a, b, C.byref(ms)); d.append(ms.value * 1e3)
        lib.bdsp_hip_event_destroy(a); lib.bdsp_hip_event_destroy(b)
    d.sort()
    return d[len(d) // 2] - event_overhead_us()


def report(name, us, units, bytes_per_unit, unit_name, us_hot=None):
    gbs = units * bytes_per_unit / us / 1e3
    row = {"config": name, "us": round(us, 2), "M%s_per_s" % unit_name: round(units / us, 1),
           "algorithmic_GBs": round(gbs, 1), "roofline_frac": round(gbs / PEAK, 4)}
    if us_hot is not None:
        row["us_input_in_cache"] = round(us_hot, 2)
        row["roofline_frac_input_in_cache"] = round(units * bytes_per_unit / us_hot / 1e3 / PEAK, 4)
    for prefix, cb in CPU.items():
        if name.startswith(prefix):
            row["cpu_baseline"] = cb
    print(json.dumps(row), flush=True)

def rnd(n, dt, k=3):
    return [torch.rand(n, device=dev, dtype=dt) * 20 - 10 for _ in range(k)]

x0 = rnd(65536, torch.float32, 1)[0]
us = timeit_fresh(lambda b, s_: (lib.bdsp_hip_dev_real_scale(0, b.data_ptr(), 65536, 2.5, sp), lib.bdsp_hip_dev_real_offset(0, b.data_ptr(), 65536, 0, -1.25, sp)), x0, 100)
report("C1 real f32 65536: scale+offset (2 launches, launch-bound)", us, 65536, 16, "samples")
n = 1 << 26
xb = rnd(n, torch.float32, 2)
us = timeit(lambda i: lib.bdsp_hip_dev_real_scale(0, xb[i % 2].data_ptr(), n, 1.0001, sp))
report("C1' real f32 64M: scale (bandwidth regime)", us, n, 8, "samples")
del xb

n = 1 << 20
sc = torch.empty(2 * n, device=dev, dtype=torch.float32); pristine = rnd(2 * n, torch.float32, 1)[0]
c2 = lambda b, s_: lib.bdsp_hip_dev_fft(0, b.data_ptr(), s_.data_ptr(), n, 1, FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp)
us = timeit_fresh(c2, pristine, 60, sc)
report("C2 complex f32 1M: plain_fft->magnitude fused (2 passes, latency-bound); every input valid, cold, used once, own scratch", us, n, 12, "points", timeit_hot(c2, pristine, 60, sc))
b = 64
sc = torch.empty(2 * n * b, device=dev, dtype=torch.float32); pristine = rnd(2 * n * b, torch.float32, 1)[0]
us = timeit_fresh(lambda bf, s_: lib.bdsp_hip_dev_fft(0, bf.data_ptr(), s_.data_ptr(), n, b, FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp), pristine, 10, sc)
report("C2x64 64 x complex f32 1M: plain_fft->magnitude fused", us, n * b, 12, "points")
del sc, pristine

n, m = 1 << 24, 1024
xs = rnd(2 * n, torch.float32); y = torch.empty(2 * n, device=dev, dtype=torch.float32)
taps = (torch.rand(2 * m, device=dev) * 2 - 1) / m
spec = torch.empty(2 * lib.bdsp_hip_conv_spectrum_points(), device=dev, dtype=torch.float32)
lib.bdsp_hip_dev_conv_prepare(0, taps.data_ptr(), m, spec.data_ptr(), sp)
us = timeit(lambda i: lib.bdsp_hip_dev_convolve_prepared(0, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, spec.data_ptr(), m, sp))
report("C3 complex f32 16M (*) 1024 taps: fused overlap-save", us, n, 16, "samples")
us = timeit(lambda i: lib.bdsp_hip_dev_convolve(0, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, sp))
report("C3 as ONE launch (taps transformed in the kernel): convolve_signal", us, n, 16, "samples")
xd = rnd(2 * n, torch.float64, 2); yd = torch.empty(2 * n, device=dev, dtype=torch.float64); td = taps.double()
us = timeit(lambda i: lib.bdsp_hip_dev_convolve(1, xd[i % 2].data_ptr(), yd.data_ptr(), n, 1, td.data_ptr(), m, sp), 10)
report("C3 in f64: complex f64 16M (*) 1024 taps", us, n, 32, "samples")
del xd, yd
f16 = lambda b, s_: lib.bdsp_hip_dev_fft(0, b.data_ptr(), s_.data_ptr(), n, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp)
us = timeit_fresh(f16, xs[0], 20, y)
report("FFT complex f32 16M: plain_fft (3 passes); every input valid, cold, used once, own scratch", us, n, 16, "points", timeit_hot(f16, xs[0], 20, y))
# real signal, real taps through the facade (B2): two real blocks per complex transform pair
import numpy as np
from basic_dsp_amd import DspVec
rv = [DspVec(np.random.rand(n).astype(np.float32) * 20 - 10) for _ in range(3)]
rh = DspVec((np.random.rand(m).astype(np.float32) * 2 - 1) / m)
import time as _tm
_tot = 0.0
for i in range(-20, 100):  # (the facade's convolve_signal works in place: every call gets a fresh clone of its input, untimed)
    w = rv[i % 3].clone()
    lib.bdsp_hip_synchronize(None)
    _t0 = _tm.perf_counter()
    w.convolve_signal(rh)
    lib.bdsp_hip_synchronize(None)
    if i >= 0: _tot += _tm.perf_counter() - _t0
    del w
report("C3 on REAL data: real f32 16M (*) 1024 real taps, through the facade (host call + synchronise included)", _tot / 100 * 1e6, n, 8, "samples")
del rv
del xs, y

n = 1 << 22
xs = rnd(2 * n, torch.float64); sc = torch.empty(2 * n, device=dev, dtype=torch.float64)
c4a = lambda b, s_: lib.bdsp_hip_dev_fft(1, b.data_ptr(), s_.data_ptr(), n, 1, FFT_SHIFT_OUT, 1.0, 4, 0.5, C.byref(flag), sp)
us = timeit_fresh(c4a, xs[0], 30, sc)
report("C4a complex f64 4M: windowed_fft(Hann) fused window+fft+shift; every input valid, cold, used once, own scratch", us, n, 32, "points", timeit_hot(c4a, xs[0], 30, sc))
# ONE protocol for C4b (round 4): three rotating inputs (64 MB each) AND three rotating outputs (256 MB each), like the
# headline's rotating inputs -- neither side of the operation finds its data in the 256 MB Infinity Cache; 30 calls
outs = [torch.empty(8 * n, device=dev, dtype=torch.float64) for _ in range(3)]
us = timeit(lambda i: lib.bdsp_hip_dev_interpolatef(1, xs[i % 3].data_ptr(), outs[i % 3].data_ptr(), 2 * n, 1, 1, 0.35, 4.0, 0.0, 12, 1.0, sp), 30)
report("C4b complex f64 4M: interpolatef(RC 0.35, x4, conv_len 12), 3 rotating inputs and outputs", us, n, 80, "input_points")
del xs, sc, outs

n, b = 1 << 20, 64
xs = rnd(2 * n * b, torch.float32, 2); y = torch.empty(2 * n * b, device=dev, dtype=torch.float32)
sc5 = torch.empty(2 * n * b, device=dev, dtype=torch.float32)
def c5(i):  # (the transform ping-pongs between the convolution's result and its OWN scratch: the inputs stay valid)
    lib.bdsp_hip_dev_convolve(0, xs[i % 2].data_ptr(), y.data_ptr(), n, b, taps.data_ptr(), m, sp)
    lib.bdsp_hip_dev_fft(0, y.data_ptr(), sc5.data_ptr(), n, b, 0, 1.0, -1, 0.0, C.byref(flag), sp)
us = timeit(c5, 10)
report("C5/GPU 64 x complex f32 1M: convolve_signal -> fft (compute only)", us, n * b, 32, "samples")
