#!/usr/bin/env python3
"""GPU box: the fractional (scalar) path of interpolatef through the device-pointer API: kernel time only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
n = 1 << 22
for dt, elem in ((torch.float32, 0), (torch.float64, 1)):
    for fid, ro, name in ((0, 0.0, "sinc"), (1, 0.35, "raised cosine")):
        xs = [torch.rand(2 * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3)]
        nl = lib.bdsp_hip_interpolatef_new_len(elem, 2 * n, 2.5)
        outs = [torch.empty(nl, device=dev, dtype=dt) for _ in range(3)]
        f = lambda i: bd._lib.check(lib.bdsp_hip_dev_interpolatef(elem, xs[i % 3].data_ptr(), outs[i % 3].data_ptr(), 2 * n, 1, fid, ro, 2.5, 0.0, 12, 1.0, sp))
        for i in range(10): f(i)
        torch.cuda.synchronize()
        e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(e0, sp)
        for i in range(10): f(i)
        lib.bdsp_hip_event_record(e1, sp)
        torch.cuda.synchronize()
        ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        b = (2 * n + nl) * (4 if elem == 0 else 8)
        print("%s %-13s factor 2.5, 4M -> 10M complex points, conv_len 12: %7.1f us  (%4.0f GB/s algorithmic)  [%s]" % (
            "f32" if elem == 0 else "f64", name, ms.value * 100, b / (ms.value * 100) / 1e3, os.path.basename(bd.LIB_PATH)))
