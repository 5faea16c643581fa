#!/usr/bin/env python3
"""GPU box: config C5's shard (64 x 1M points): convolve all -> transform all, against chunks of K vectors taken through
convolve -> transform one after the other (the chunk's convolution result is then still in the Infinity Cache)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
n, b, m = 1 << 20, 64, 1024
xs = [torch.rand(2 * n * b, device=dev) * 20 - 10 for _ in range(2)]
y = torch.empty(2 * n * b, device=dev)
s = torch.empty(2 * n * b, device=dev)
taps = (torch.rand(2 * m, device=dev) * 2 - 1) / m
spec = torch.empty(2 * lib.bdsp_hip_conv_spectrum_points(), device=dev)
lib.bdsp_hip_dev_conv_prepare(0, taps.data_ptr(), m, spec.data_ptr(), sp)
E = 8  # bytes per complex f32


def run(i, k):
    x = xs[i % 2]
    for c0 in range(0, b, k):
        off = c0 * n * E
        bd._lib.check(lib.bdsp_hip_dev_convolve_prepared(0, x.data_ptr() + off, y.data_ptr() + off, n, k, spec.data_ptr(), m, sp))
        bd._lib.check(lib.bdsp_hip_dev_fft(0, y.data_ptr() + off, s.data_ptr() + off, n, k, 0, 1.0, -1, 0.0, C.byref(flag), sp))


def timed(k, iters=20):
    import time
    t0 = time.perf_counter(); j = 0
    while time.perf_counter() - t0 < 0.3:
        run(j, k); j += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): run(i, k)
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3


for rep in range(2):
    for k in (64, 32, 16, 8, 4):
        print("chunks of %2d vectors: %.1f us per shard" % (k, timed(k))); sys.stdout.flush()
