#!/usr/bin/env python3
"""GPU box: config C4b's time right after start-up and after seconds of sustained load (tools/bench_configs.py measures it
after ten seconds of other configs and prints 54-56 us; a fresh process prints 68)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
n = 1 << 22
dt = torch.float64
xs = [torch.rand(2 * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3)]
outs = [torch.empty(8 * n, device=dev, dtype=dt) for _ in range(3)]
flag = C.c_int(0)


def c4b(i): lib.bdsp_hip_dev_interpolatef(1, xs[i % 3].data_ptr(), outs[i % 3].data_ptr(), 2 * n, 1, 1, 0.35, 4.0, 0.0, 12, 1.0, sp)


def timed(iters=30):
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): c4b(i)
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3


c4b(0); torch.cuda.synchronize()
print("right after start-up:            %.1f us" % timed())
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 0.15:
    for _ in range(10): c4b(k); k += 1
    torch.cuda.synchronize()
print("after the usual 0.15 s pre-warm: %.1f us" % timed())
for secs in (1, 3, 6):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(50): c4b(k); k += 1
        torch.cuda.synchronize()
    print("after %d more s of itself:        %.1f us, %.1f us" % (secs, timed(), timed(100)))
big = torch.rand(1 << 27, device=dev)
sc = torch.empty(1 << 27, device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 4:
    for _ in range(20): lib.bdsp_hip_dev_fft(0, big.data_ptr(), sc.data_ptr(), 1 << 26, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp)
    torch.cuda.synchronize()
print("after 4 s of 64M-point f32 FFTs:  %.1f us, %.1f us" % (timed(), timed(100)))
time.sleep(2)
print("after 2 s of idle:               %.1f us" % timed())
