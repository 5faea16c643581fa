#!/usr/bin/env python3
"""Quick per-kernel timing on the GPU box: conv (prepared spectrum) and FFT at a given size."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1 << 24)
ap.add_argument("--taps", type=int, default=1024)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--elem", type=int, default=0)
ap.add_argument("--what", default="conv,fft")
ap.add_argument("--zero", action="store_true", help="all-zero input (shows how much of a difference is the clock the data draws)")
a = ap.parse_args()
n, m, b = a.points, a.taps, a.batch
dt = torch.float32 if a.elem == 0 else torch.float64
dev = torch.device("cuda", 0)
xs = [(torch.zeros(2 * n * b, device=dev, dtype=dt) if a.zero else torch.rand(2 * n * b, device=dev, dtype=dt) * 20 - 10) for _ in range(3)]
taps = (torch.rand(2 * m, device=dev, dtype=dt) * 2 - 1) / m
y = torch.empty(2 * n * b, device=dev, dtype=dt)
spec = torch.empty(2 * lib.bdsp_hip_conv_spectrum_points(), device=dev, dtype=dt)
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
def timeit(fn):
    # untimed pre-warm: the clock needs tens of milliseconds of load to settle (see bench.py)
    import time as _t
    t0 = _t.perf_counter(); k = 0
    while _t.perf_counter() - t0 < 0.15:
        for _ in range(10): fn(k); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(a.iters): fn(i)
    lib.bdsp_hip_event_record(e1, sp)
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / a.iters * 1e3
esz = 8 if a.elem == 0 else 16
if "conv" in a.what.split(",") and m > 3073:
    us = timeit(lambda i: bd._lib.check(lib.bdsp_hip_dev_convolve(a.elem, xs[i % 3].data_ptr(), y.data_ptr(), n, b, taps.data_ptr(), m, sp)))
    print("conv  n=%d b=%d m=%d (long-filter path): %.1f us  %.1f Gsamples/s" % (n, b, m, us, n * b / us / 1e3))
elif "conv" in a.what.split(","):
    bd._lib.check(lib.bdsp_hip_dev_conv_prepare(a.elem, taps.data_ptr(), m, spec.data_ptr(), sp))
    us = timeit(lambda i: bd._lib.check(lib.bdsp_hip_dev_convolve_prepared(a.elem, xs[i % 3].data_ptr(), y.data_ptr(), n, b, spec.data_ptr(), m, sp)))
    print("conv  n=%d b=%d m=%d: %.1f us  %.1f Gsamples/s  %.0f GB/s algorithmic" % (n, b, m, us, n * b / us / 1e3, 2 * esz * n * b / us / 1e3))
if "prepconv" in a.what:
    def f(i):
        bd._lib.check(lib.bdsp_hip_dev_conv_prepare(a.elem, taps.data_ptr(), m, spec.data_ptr(), sp))
        bd._lib.check(lib.bdsp_hip_dev_convolve_prepared(a.elem, xs[i % 3].data_ptr(), y.data_ptr(), n, b, spec.data_ptr(), m, sp))
    us = timeit(f)
    print("prep+conv n=%d m=%d: %.1f us" % (n, m, us))
if "convfft" in a.what:
    scr = torch.empty_like(y)
    def g(i):
        bd._lib.check(lib.bdsp_hip_dev_convolve_prepared(a.elem, xs[i % 3].data_ptr(), y.data_ptr(), n, b, spec.data_ptr(), m, sp))
        bd._lib.check(lib.bdsp_hip_dev_fft(a.elem, y.data_ptr(), scr.data_ptr(), n, b, 0, 1.0, -1, 0.0, C.byref(flag), sp))
    bd._lib.check(lib.bdsp_hip_dev_conv_prepare(a.elem, taps.data_ptr(), m, spec.data_ptr(), sp))
    us = timeit(g)
    print("conv+fft (no prepare) n=%d m=%d: %.1f us" % (n, m, us))
if "fft" in a.what.split(","):
    # (the transform clobbers its input: every timed call gets its own copy of valid data, used once -- a loop over the same
    # three buffers would run on inf / NaN after a few dozen calls)
    fresh = [xs[0].clone() for _ in range(min(a.iters, max(3, int(24e9 // (xs[0].numel() * xs[0].element_size())))))]
    a.iters = len(fresh)
    fftc = lambda buf: bd._lib.check(lib.bdsp_hip_dev_fft(a.elem, buf.data_ptr(), y.data_ptr(), n, b, 0, 1.0, -1, 0.0, C.byref(flag), sp))
    import time as _t
    t0 = _t.perf_counter(); k = 0
    while _t.perf_counter() - t0 < 0.15:  # pre-warm on the three scratch inputs (their contents do not matter)
        for _ in range(10): fftc(xs[k % 3]); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for buf in fresh: fftc(buf)
    lib.bdsp_hip_event_record(e1, sp)
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    us = ms.value / len(fresh) * 1e3
    print("fft   n=%d b=%d: %.1f us  %.1f Gsamples/s  %.0f GB/s algorithmic" % (n, b, us, n * b / us / 1e3, 2 * esz * n * b / us / 1e3))
if "tapsconv" in a.what.split(","):
    us = timeit(lambda i: bd._lib.check(lib.bdsp_hip_dev_convolve(a.elem, xs[i % 3].data_ptr(), y.data_ptr(), n, b, taps.data_ptr(), m, sp)))
    print("conv (taps transformed in the kernel) n=%d b=%d m=%d: %.1f us" % (n, b, m, us))
if "prep" in a.what.split(","):
    us = timeit(lambda i: bd._lib.check(lib.bdsp_hip_dev_conv_prepare(a.elem, taps.data_ptr(), m, spec.data_ptr(), sp)))
    print("prep  m=%d: %.1f us" % (m, us))
