#!/usr/bin/env python3
"""GPU box: does config C4b's time depend on WHERE its buffers lie?  Same call, same protocol (three rotating inputs and
outputs), different allocation orders / paddings; prints the device addresses modulo a few powers of two."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
n = 1 << 22
dt = torch.float64


def timeit(fn, iters=30):
    import time
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 0.15:
        for _ in range(10): fn(k); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): fn(i)
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3


def run(label, xs, outs):
    us = timeit(lambda i: lib.bdsp_hip_dev_interpolatef(1, xs[i % 3].data_ptr(), outs[i % 3].data_ptr(), 2 * n, 1, 1, 0.35, 4.0, 0.0, 12, 1.0, sp))
    a = ["%x/%x" % (x.data_ptr() >> 21 & 0xfff, o.data_ptr() >> 21 & 0xfff) for x, o in zip(xs, outs)]
    print("%-64s %.1f us   2MB-page indices in/out: %s" % (label, us, " ".join(a))); sys.stdout.flush()


mk_in = lambda: [torch.rand(2 * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3)]
mk_out = lambda: [torch.empty(8 * n, device=dev, dtype=dt) for _ in range(3)]
xs = mk_in(); outs = mk_out()
run("inputs allocated first, then outputs (fresh process)", xs, outs)
run("the same buffers again", xs, outs)
del xs, outs; torch.cuda.empty_cache()
outs = mk_out(); xs = mk_in()
run("outputs first, then inputs", xs, outs)
del xs, outs; torch.cuda.empty_cache()
pad = torch.empty(37 * (1 << 20) + 4096, device=dev, dtype=torch.uint8)
xs = mk_in(); pad2 = torch.empty(113 * (1 << 20), device=dev, dtype=torch.uint8); outs = mk_out()
run("odd paddings before and between", xs, outs)
del xs, outs; torch.cuda.empty_cache()
big = torch.empty(3 * 2 * n + 3 * 8 * n + (1 << 20), device=dev, dtype=dt)
xs = [big[i * 2 * n:(i + 1) * 2 * n].uniform_(-10, 10) for i in range(3)]
off = 3 * 2 * n + 512
outs = [big[off + i * 8 * n: off + (i + 1) * 8 * n] for i in range(3)]
run("one big allocation, inputs then outputs back to back (+4 KB)", xs, outs)
# like tools/bench_configs.py: other work and allocations before
del xs, outs, big; torch.cuda.empty_cache()
junk = [torch.rand(1 << 26, device=dev) for _ in range(2)]
for _ in range(200): junk[0].mul_(1.0001)
xs = mk_in(); sc = torch.empty(2 * n, device=dev, dtype=dt); outs = mk_out()
run("after other allocations and work (bench_configs-like)", xs, outs)
