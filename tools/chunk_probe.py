#!/usr/bin/env python3
"""Does scheduling a batch in Infinity-Cache-sized chunks help?  64 x 1M-point complex f32 plain_fft -> magnitude through the
B3 device API (data -> scratch -> data), whole batch in one call against chunks of k vectors that share ONE k * 8 MB scratch
buffer, so the intermediate can stay in the 256 MB cache.  Run on the GPU box: python tools/chunk_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
from basic_dsp_amd._lib import FFT_MAGNITUDE
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
n, b = 1 << 20, 64
xs = [torch.rand(2 * n * b, device=dev) * 20 - 10 for _ in range(2)]
scratch = torch.empty(2 * n * b, device=dev)

def run(i, k):
    x = xs[i % 2]
    for c in range(0, b, k):
        lib.bdsp_hip_dev_fft(0, x.data_ptr() + c * n * 8, scratch.data_ptr(), n, min(k, b - c), FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp)

def timeit(fn, iters=10):
    import time
    t0 = time.perf_counter(); j = 0
    while time.perf_counter() - t0 < 0.15:
        fn(j); j += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): fn(i)
    lib.bdsp_hip_event_record(e1, sp)
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3

ref = None
for k in (64, 32, 24, 20, 16, 12, 10, 8, 4):
    us = timeit(lambda i: run(i, k))
    xs[0].uniform_(-10, 10, generator=torch.Generator(device=dev).manual_seed(7))
    run(0, k); torch.cuda.synchronize()
    chk = sum(float(xs[0][2 * n * c: 2 * n * c + n * min(k, b - c)].double().sum()) for c in range(0, b, k))  # compact per call
    if ref is None: ref = chk
    print("chunks of %2d vectors: %7.1f us per 64-vector batch   (checksum ratio %.12f)" % (k, us, chk / ref))
