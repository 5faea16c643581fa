#!/usr/bin/env python3
"""Times and CHECKS alternative super-radix plans (BDSP_FFT_PLAN) for one length and precision.
usage: BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab.so BDSP_FFT_PLAN=2048x4,2048x4 python tools/plan_probe.py <log2 n> <f32|f64>
(the plan switch exists only in the LAB build of the library: make -C basic_dsp_amd/csrc lab)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import basic_dsp_amd as bd
lib = bd.lib
sp = bd._lib.torch_stream_arg()
bits, prec = int(sys.argv[1]), sys.argv[2]
n = 1 << bits
dt = torch.float32 if prec == "f32" else torch.float64
elem = 0 if prec == "f32" else 1
g = torch.Generator(device="cuda").manual_seed(5)
xs = [torch.rand(2 * n, device="cuda", dtype=dt, generator=g) * 20 - 10 for _ in range(3)]
keep = xs[0].clone()
y = torch.empty(2 * n, device="cuda", dtype=dt)
flag = C.c_int(0)
def run(i): return lib.bdsp_hip_dev_fft(elem, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp)
rc = run(0); torch.cuda.synchronize()
if rc != 0:
    print("plan %-24s 2^%d %s: rc=%d (unsupported)" % (os.environ.get("BDSP_FFT_PLAN", "default"), bits, prec, rc)); sys.exit(0)
res = (y if flag.value else xs[0]).double().cpu().numpy().view(np.complex128)
ref = np.fft.fft(keep.double().cpu().numpy().view(np.complex128))
err = np.linalg.norm(res - ref) / np.linalg.norm(ref)
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 0.15:
    for _ in range(10): run(k); k += 1
    torch.cuda.synchronize()
e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
lib.bdsp_hip_event_record(e0, sp)
for i in range(30): run(i)
lib.bdsp_hip_event_record(e1, sp)
ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
print("plan %-24s 2^%d %s: %7.1f us   rel-L2 vs numpy %.2e" % (os.environ.get("BDSP_FFT_PLAN", "default"), bits, prec, ms.value / 30 * 1e3, err))
