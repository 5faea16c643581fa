#!/usr/bin/env python3
"""GPU box: the 16M-point transform (and the bench step) on torch's default stream (= HIP's null stream) against a
created stream; one event pair around N back-to-back calls."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
flag = C.c_int(0)
n, m = 1 << 24, 1024
xs = [torch.rand(2 * n, device=dev) * 20 - 10 for _ in range(3)]
taps = (torch.rand(2 * m, device=dev) * 2 - 1) / m
y = torch.rand(2 * n, device=dev)
s = torch.empty(2 * n, device=dev)
torch.cuda.synchronize()


def run(label, sp, sync):
    def fft(): bd._lib.check(lib.bdsp_hip_dev_fft(0, y.data_ptr(), s.data_ptr(), n, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp))
    def conv(i): bd._lib.check(lib.bdsp_hip_dev_convolve(0, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, sp))
    for what in ("fft", "step"):
        for i in range(1500):
            if what == "step": conv(i)
            fft()
        sync()
        res = []
        for rep in range(3):
            e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
            lib.bdsp_hip_event_record(e0, sp)
            for i in range(100):
                if what == "step": conv(i)
                fft()
            lib.bdsp_hip_event_record(e1, sp)
            sync()
            ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
            res.append(ms.value * 10)
        print("%-44s %-5s %s us per call" % (label, what, " ".join("%.1f" % r for r in res))); sys.stdout.flush()


run("torch default stream (HIP null stream)", bd._lib.torch_stream_arg(), torch.cuda.synchronize)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    run("torch.cuda.Stream() (created, blocking)", bd._lib.torch_stream_arg(), torch.cuda.synchronize)
run("the library's own stream (non-blocking)", None, lambda: lib.bdsp_hip_synchronize(None))
