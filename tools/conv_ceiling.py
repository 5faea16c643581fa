#!/usr/bin/env python3
"""Ceilings of the f64 and real-data block kernels from timing-only ablations of the product kernel (LAB library:
make -C basic_dsp_amd/csrc lab; BDSP_CONV_ABL = 3: no global loads / stores = arithmetic + LDS exchanges alone,
8: no transform = the kernel's own memory skeleton).  Run on the GPU box:
    BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab.so python tools/conv_ceiling.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec
lib = bd.lib
assert "lab" in os.path.basename(bd.LIB_PATH), "needs the LAB library (BDSP_HIP_LIBRARY)"
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
n, m = 1 << 24, 1024


def timed(fn, iters):
    import time
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10): fn(k); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): fn(i)
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3


def sweep(name, fn, iters, algo_bytes):
    row = {"kernel": name}
    for label, abl in (("full", None), ("arithmetic_and_exchanges_only", "3"), ("memory_skeleton_only", "8")):
        if abl is None: os.environ.pop("BDSP_CONV_ABL", None)
        else: os.environ["BDSP_CONV_ABL"] = abl
        us = timed(fn, iters)
        row[label + "_us"] = round(us, 1)
        if label != "arithmetic_and_exchanges_only":
            row[label + "_frac_of_8TBs"] = round(algo_bytes / us / 1e3 / 8000.0, 3)
    os.environ.pop("BDSP_CONV_ABL", None)
    print(json.dumps(row)); sys.stdout.flush()


for dt, elem, nm in ((torch.float32, 0, "complex f32"), (torch.float64, 1, "complex f64")):
    xs = [torch.rand(2 * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3 if elem == 0 else 2)]
    y = torch.empty(2 * n, device=dev, dtype=dt)
    taps = ((torch.rand(2 * m, device=dev, dtype=dt) * 2 - 1) / m)
    sweep("k_overlap_save_v2 %s 16M (*) 1024 taps" % nm,
          lambda i: bd._lib.check(lib.bdsp_hip_dev_convolve(elem, xs[i % len(xs)].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, sp)),
          30 if elem == 0 else 15, 2 * n * (8 if elem == 0 else 16))
    del xs, y
# real data through the facade (B2), library stream
rv = [DspVec(np.random.rand(n).astype(np.float32) * 20 - 10) for _ in range(3)]
rh = DspVec((np.random.rand(m).astype(np.float32) * 2 - 1) / m)
sp = None


def real_conv(i): rv[i % 3].convolve_signal(rh)


def timed_host(fn, iters):
    import time
    for i in range(50): fn(i)
    lib.bdsp_hip_synchronize(None)
    t0 = time.perf_counter()
    for i in range(iters): fn(i)
    lib.bdsp_hip_synchronize(None)
    return (time.perf_counter() - t0) / iters * 1e6


row = {"kernel": "k_overlap_save_v2<REAL> real f32 16M (*) 1024 real taps (facade call, host wall clock)"}
for label, abl in (("full", None), ("arithmetic_and_exchanges_only", "3"), ("memory_skeleton_only", "8")):
    if abl is None: os.environ.pop("BDSP_CONV_ABL", None)
    else: os.environ["BDSP_CONV_ABL"] = abl
    us = timed_host(real_conv, 200)
    row[label + "_us"] = round(us, 1)
    if label != "arithmetic_and_exchanges_only":
        row[label + "_frac_of_8TBs"] = round(8.0 * n / us / 1e3 / 8000.0, 3)
os.environ.pop("BDSP_CONV_ABL", None)
print(json.dumps(row))
