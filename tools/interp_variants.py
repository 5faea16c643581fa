#!/usr/bin/env python3
"""GPU box: interpolatef through the device-pointer API (no facade reallocation), three rotating inputs and outputs:
every dtype / complex / factor variant against its algorithmic bytes."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
n = 1 << 22


def timeit(fn, iters=20):
    import time
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 0.15:
        for _ in range(5): fn(k); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): fn(i)
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3


for dt, elem in ((torch.float32, 0), (torch.float64, 1)):
    for cplx in (1, 0):
        for factor in (2.0, 4.0, 8.0):
            e = 2 if cplx else 1
            xs = [torch.rand(e * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3)]
            outs = [torch.empty(int(e * n * factor), device=dev, dtype=dt) for _ in range(3)]
            us = timeit(lambda i: bd._lib.check(lib.bdsp_hip_dev_interpolatef(elem, xs[i % 3].data_ptr(), outs[i % 3].data_ptr(), e * n, cplx, 1, 0.35, factor, 0.0, 12, 1.0, sp)))
            b = e * n * (1 + factor) * (4 if elem == 0 else 8)
            print(json.dumps({"variant": "%s %s 4M x%d conv_len 12" % ("f32" if elem == 0 else "f64", "complex" if cplx else "real", int(factor)),
                              "us": round(us, 1), "GBs": round(b / us / 1e3), "frac_of_8TBs": round(b / us / 1e3 / 8000, 3)}))
            sys.stdout.flush()
            del xs, outs
# the C4b protocol of tools/bench_configs.py in this process, for comparison
dt, elem, cplx, factor, e = torch.float64, 1, 1, 4.0, 2
xs = [torch.rand(e * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3)]
outs = [torch.empty(8 * n, device=dev, dtype=dt) for _ in range(3)]
for it in (10, 30, 100):
    us = timeit(lambda i: lib.bdsp_hip_dev_interpolatef(1, xs[i % 3].data_ptr(), outs[i % 3].data_ptr(), 2 * n, 1, 1, 0.35, 4.0, 0.0, 12, 1.0, sp), it)
    print("C4b again, %d iterations: %.1f us" % (it, us))
