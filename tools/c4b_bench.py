#!/usr/bin/env python3
"""Config C4b (and its siblings) through the B3 device API: interpolatef x4, RC 0.35, conv_len 12, 4M input points.
usage: [BDSP_HIP_LIBRARY=...] python tools/c4b_bench.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
n = 1 << 22
for elem, dt in ((1, torch.float64), (0, torch.float32)):
    for cplx in (1, 0):
        e = 2 if cplx else 1
        xs = [torch.rand(e * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3)]
        out = torch.empty(4 * e * n, device=dev, dtype=dt)
        def f(i):
            bd._lib.check(lib.bdsp_hip_dev_interpolatef(elem, xs[i % 3].data_ptr(), out.data_ptr(), e * n, cplx, 1, 0.35, 4.0, 0.0, 12, 1.0, sp))
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 0.15:
            for _ in range(10): f(k); k += 1
            torch.cuda.synchronize()
        e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(e0, sp)
        for i in range(100): f(i)
        lib.bdsp_hip_event_record(e1, sp)
        torch.cuda.synchronize()
        ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        us = ms.value / 100 * 1e3
        by = 5 * e * n * (8 if elem else 4)
        print("%s %s 4M -> 16M: %.1f us  %.0f GB/s algorithmic = %.3f of 8 TB/s" % ("f64" if elem else "f32", "complex" if cplx else "real", us, by / us / 1e3, by / us / 1e3 / 8000))
        del xs, out
