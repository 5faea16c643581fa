#!/usr/bin/env python3
"""GPU box: does running the NEXT vector's convolution beside the current vector's transform (two streams, independent
vectors, double-buffered y) beat the one-stream step?  Wall clock over 300 steps, everything resident."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
flag = C.c_int(0)
n, m = 1 << 24, 1024
xs = [torch.rand(2 * n, device=dev) * 20 - 10 for _ in range(3)]
taps = (torch.rand(2 * m, device=dev) * 2 - 1) / m
ys = [torch.empty(2 * n, device=dev) for _ in range(2)]
ss = [torch.empty(2 * n, device=dev) for _ in range(2)]
st = [torch.cuda.Stream(device=dev) for _ in range(2)]
sps = [bd._lib.stream_arg(s.cuda_stream) for s in st]
torch.cuda.synchronize()


def step(i, k, sp):
    bd._lib.check(lib.bdsp_hip_dev_convolve(0, xs[i % 3].data_ptr(), ys[k].data_ptr(), n, 1, taps.data_ptr(), m, sp))
    bd._lib.check(lib.bdsp_hip_dev_fft(0, ys[k].data_ptr(), ss[k].data_ptr(), n, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp))


def timed(label, fn, steps=300):
    for i in range(1500): fn(i)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(steps): fn(i)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e6)
    print("%-72s %s us per step" % (label, " ".join("%.1f" % r for r in res))); sys.stdout.flush()


timed("one stream, one (y, s) pair (the bench step)", lambda i: step(i, 0, sps[0]))
timed("one stream, alternating (y, s) pairs (512 MB of intermediates)", lambda i: step(i, i & 1, sps[0]))
timed("two streams, steps alternate between them (independent vectors overlap)", lambda i: step(i, i & 1, sps[i & 1]))
