#!/usr/bin/env python3
"""interpolatef timing (config C4b and f32 / real variants)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as orc
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec
for dtype in (np.float64, np.float32):
    for cplx in (True, False):
        n = 1 << 22
        e = 2 if cplx else 1
        x = orc.fill_uniform(n * e, 1, -10, 10, dtype)
        vs = [DspVec(x, is_complex=cplx) for _ in range(6)]
        vs[0].interpolatef(1, 4.0, 0.0, 12, rolloff=0.35)
        bd.lib.bdsp_hip_synchronize(None)
        t0 = time.perf_counter()
        for v in vs[1:]:
            v.interpolatef(1, 4.0, 0.0, 12, rolloff=0.35)
        bd.lib.bdsp_hip_synchronize(None)
        us = (time.perf_counter() - t0) / 5 * 1e6
        by = n * e * np.dtype(dtype).itemsize * 5
        print("%s %s 4M -> 16M: %.1f us  (%.0f GB/s algorithmic)" % (np.dtype(dtype).name, "complex" if cplx else "real", us, by / us / 1e3))
