#!/usr/bin/env python3
"""GPU box: the half-column plan for ONE 2^20-point f32 vector (k_fft_half512) against numpy in f64 for every output
option it serves, and its time against the 1024 x 4 tiles (LAB library: BDSP_FFT_H512 unset).
usage: BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab.so python tools/c2_half512_check.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import basic_dsp_amd as bd
from basic_dsp_amd._lib import FFT_SHIFT_OUT, FFT_MAGNITUDE
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
n = 1 << 20
rng = np.random.default_rng(3)
x = (rng.random(2 * n, dtype=np.float32) * 20 - 10)
xc = x[0::2].astype(np.float64) + 1j * x[1::2].astype(np.float64)
ref = np.fft.fft(xc)
refi = np.fft.ifft(xc) * n


def run(flags):
    d = torch.from_numpy(x).to(dev)
    s = torch.empty(2 * n, device=dev, dtype=torch.float32)
    bd._lib.check(lib.bdsp_hip_dev_fft(0, d.data_ptr(), s.data_ptr(), n, 1, flags, 1.0, -1, 0.0, C.byref(flag), sp))
    torch.cuda.synchronize()
    return (s if flag.value else d).cpu().numpy()


def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))


o = run(0); print("plain_fft        rel-L2 %.3e" % rel(o[0::2] + 1j * o[1::2], ref))
o = run(1); print("inverse          rel-L2 %.3e" % rel(o[0::2] + 1j * o[1::2], refi))
o = run(FFT_SHIFT_OUT); print("fft (shift)      rel-L2 %.3e" % rel(o[0::2] + 1j * o[1::2], np.fft.fftshift(ref)))
o = run(FFT_MAGNITUDE); print("fft->magnitude   rel-L2 %.3e" % rel(o[:n], np.abs(ref)))
o = run(FFT_MAGNITUDE | FFT_SHIFT_OUT); print("fft shift->magn. rel-L2 %.3e" % rel(o[:n], np.abs(np.fft.fftshift(ref))))


def timeit(fn, iters=200):
    """Every call gets its input restored first (untimed copy; the transform clobbers it -- a loop of in-place transforms
    otherwise runs on inf / NaN after a few dozen calls) and carries its own event pair."""
    import time
    for k in range(300):
        xs[(k + 1) % 3].copy_(pristine); fn(k)
    torch.cuda.synchronize()
    ov = []
    ms = C.c_float(0)
    for _ in range(20):
        a, b = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(a, sp); lib.bdsp_hip_event_record(b, sp)
        lib.bdsp_hip_event_elapsed_ms(a, b, C.byref(ms)); ov.append(ms.value * 1e3)
    pairs = []
    for i in range(iters):
        xs[(i + 1) % 3].copy_(pristine)
        a, b = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(a, sp); fn(i); lib.bdsp_hip_event_record(b, sp)
        pairs.append((a, b))
    torch.cuda.synchronize()
    d = []
    for a, b in pairs:
        lib.bdsp_hip_event_elapsed_ms(a, b, C.byref(ms)); d.append(ms.value * 1e3)
    d.sort()
    return d[len(d) // 2] - min(ov)


pristine = torch.rand(2 * n, device=dev, dtype=torch.float32) * 20 - 10
xs = [pristine.clone() for _ in range(3)]
sc = torch.empty(2 * n, device=dev, dtype=torch.float32)
def timeit_cold(flags, iters=200):
    """every call on its own copy of the input, used once: a cold 8 MB read per call; one event pair around the loop"""
    bufs = [pristine.clone() for _ in range(iters)]
    for k in range(300):
        xs[k % 3].copy_(pristine); lib.bdsp_hip_dev_fft(0, xs[k % 3].data_ptr(), sc.data_ptr(), n, 1, flags, 1.0, -1, 0.0, C.byref(flag), sp)
    torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for b_ in bufs: lib.bdsp_hip_dev_fft(0, b_.data_ptr(), sc.data_ptr(), n, 1, flags, 1.0, -1, 0.0, C.byref(flag), sp)
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / iters * 1e3


print("C2 COLD input: plain_fft->magnitude %.2f us, plain_fft %.2f us" % (timeit_cold(FFT_MAGNITUDE), timeit_cold(0)))
us = timeit(lambda i: lib.bdsp_hip_dev_fft(0, xs[i % 3].data_ptr(), sc.data_ptr(), n, 1, FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp))
print("C2 plain_fft->magnitude: %.2f us   (%s%s)" % (us, os.path.basename(bd.LIB_PATH), ", BDSP_FFT_H512" if os.environ.get("BDSP_FFT_H512") else ""))
us = timeit(lambda i: lib.bdsp_hip_dev_fft(0, xs[i % 3].data_ptr(), sc.data_ptr(), n, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp))
print("   plain_fft           : %.2f us" % us)
