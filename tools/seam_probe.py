#!/usr/bin/env python3
"""GPU box: what does the 16M-point transform cost depending on what ran before it?  (The transform alone on two hot
buffers takes ~103 us, inside the bench step ~120 us.)  HIP events around the transform only, 60 repetitions each."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
n, m = 1 << 24, 1024
xs = [torch.rand(2 * n, device=dev) * 20 - 10 for _ in range(3)]
taps = (torch.rand(2 * m, device=dev) * 2 - 1) / m
y = torch.rand(2 * n, device=dev)
s = torch.empty(2 * n, device=dev)
junk = torch.empty(2 * n, device=dev)
small = torch.zeros(1 << 16, device=dev)


def fft(): bd._lib.check(lib.bdsp_hip_dev_fft(0, y.data_ptr(), s.data_ptr(), n, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp))
def conv(i): bd._lib.check(lib.bdsp_hip_dev_convolve(0, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, sp))


def measure(name, before, reps=60):
    for i in range(300):  # clock
        conv(i); fft()
    torch.cuda.synchronize()
    ev = [(lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()) for _ in range(reps)]
    for i in range(reps):
        before(i)
        lib.bdsp_hip_event_record(ev[i][0], sp)
        fft()
        lib.bdsp_hip_event_record(ev[i][1], sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); t = []
    for a, b in ev:
        lib.bdsp_hip_event_elapsed_ms(a, b, C.byref(ms)); t.append(ms.value * 1e3)
        lib.bdsp_hip_event_destroy(a); lib.bdsp_hip_event_destroy(b)
    t.sort()
    print("%-78s median %6.1f us  (min %6.1f, max %6.1f)" % (name, t[len(t) // 2], t[0], t[-1])); sys.stdout.flush()


measure("transform alone, back to back on (y, s)", lambda i: None)
measure("after convolve_signal(x_i -> y), three rotating inputs (the bench step)", conv)
measure("after convolve_signal(x_0 -> y), one input", lambda i: conv(0))
measure("after a 128 MB READ of x_i (torch sum), y untouched", lambda i: xs[i % 3].sum())
measure("after a 128 MB WRITE to a junk buffer (torch fill_), y untouched", lambda i: junk.fill_(1.0))
measure("after a 128 MB copy x_i -> junk (256 MB of traffic), y untouched", lambda i: junk.copy_(xs[i % 3]))
measure("after a 128 MB copy x_i -> y (y freshly written, no arithmetic)", lambda i: y.copy_(xs[i % 3]))
def conv_then_rest(i):
    conv(i)
    for _ in range(40): small.add_(1.0)   # ~100+ us of near-idle launches
measure("after convolve_signal(x_i -> y) and ~40 tiny kernels of rest", conv_then_rest)
