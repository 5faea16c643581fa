#!/usr/bin/env python3
"""Round 6's new kernels on valid data, a few launches each, for rocprofv3 (tools/profile_new_kernels.sh): k_mr_reg3 (16384 x 1000
and 4096 x 3000 points f32, 16384 x 1000 f64), k_mr_reg2 (65536 x 100 points f32), k_interp_frac_pk (4M -> 10M complex points, factor
2.5, conv_len 12: sinc and raised cosine, f32 and f64).  Every transform runs on a buffer that was restored from a pristine copy of
the random input just before it (valid data, input in the caches: the "hot" protocol of tools/plan_probe.py), forty times per
configuration so that the kernel trace's average is taken at the sustained clock, not on the ramp."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
REPS = 40
for n, batch, dt, elem in ((1000, 16384, torch.float32, 0), (3000, 4096, torch.float32, 0), (1000, 16384, torch.float64, 1), (100, 65536, torch.float32, 0)):
    pristine = torch.rand(2 * n * batch, device=dev, dtype=dt) * 20 - 10
    bufs = [pristine.clone() for _ in range(3)]
    scr = torch.empty_like(pristine)
    for i in range(REPS):
        bufs[i % 3].copy_(pristine)
        bd._lib.check(lib.bdsp_hip_dev_fft(elem, bufs[i % 3].data_ptr(), scr.data_ptr(), n, batch, 0, 1.0, -1, 0.0, C.byref(flag), sp))
    torch.cuda.synchronize()
    del bufs, scr, pristine
n = 1 << 22
for dt, elem in ((torch.float32, 0), (torch.float64, 1)):
    for fid, ro in ((0, 0.0), (1, 0.35)):
        xs = [torch.rand(2 * n, device=dev, dtype=dt) * 20 - 10 for _ in range(3)]
        nl = lib.bdsp_hip_interpolatef_new_len(elem, 2 * n, 2.5)
        outs = [torch.empty(nl, device=dev, dtype=dt) for _ in range(3)]
        for i in range(12):
            bd._lib.check(lib.bdsp_hip_dev_interpolatef(elem, xs[i % 3].data_ptr(), outs[i % 3].data_ptr(), 2 * n, 1, fid, ro, 2.5, 0.0, 12, 1.0, sp))
        torch.cuda.synchronize()
        del xs, outs
print("done")
