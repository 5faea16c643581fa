#!/usr/bin/env python3
"""Scans the block kernel's dispatch-group shares (per call, bdsp_hip_dev_convolve_ex) on the box: 16M points x 1024 taps,
f32 complex.  usage: python tools/share_scan.py [points] [taps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
xs = [torch.rand(2 * n, device=dev) * 20 - 10 for _ in range(3)]
taps = (torch.rand(2 * m, device=dev) * 2 - 1) / m
y = torch.empty(2 * n, device=dev)
sp = bd._lib.torch_stream_arg()
def run(i, sh=(-1, -1)):
    bd._lib.check(lib.bdsp_hip_dev_convolve_ex(0, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, sh[0], sh[1], sp))
def timed(reps, sh=(-1, -1)):
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(reps): run(i, sh)
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / reps * 1e3
for i in range(3000): run(i)
torch.cuda.synchronize()
res = {}
for rep in range(2):
    for a in (39, 41, 43, 45, 47):
        for b in (33, 35, 37, 39):
            if a + b > 90: continue
            res.setdefault((a, b), []).append(timed(300, (a, b)))
    res.setdefault("default", []).append(timed(300))
for k, v in sorted(res.items(), key=lambda kv: min(kv[1])):
    print(k, " ".join("%.2f" % t for t in v))
