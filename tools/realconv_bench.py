#!/usr/bin/env python3
"""Real-signal convolve_signal timing through the facade (device-resident)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as orc
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec
for dtype in (np.float32, np.float64):
    for cplx in (False, True):
        n, m = 1 << 24, 1024
        e = 2 if cplx else 1
        x = orc.fill_uniform(n * e, 1, -10, 10, dtype)
        h = orc.fill_uniform(m * e, 2, -1, 1, dtype)
        v, hv = DspVec(x, is_complex=cplx), DspVec(h, is_complex=cplx)
        for _ in range(3): v.convolve_signal(hv)
        bd.lib.bdsp_hip_synchronize(None)
        t0 = time.perf_counter()
        for _ in range(20): v.convolve_signal(hv)
        bd.lib.bdsp_hip_synchronize(None)
        us = (time.perf_counter() - t0) / 20 * 1e6
        print("%s %s 16M x 1024 taps: %.1f us  (%.0f GB/s algorithmic)" % (np.dtype(dtype).name, "complex" if cplx else "real", us, 2 * n * e * np.dtype(dtype).itemsize / us / 1e3))
