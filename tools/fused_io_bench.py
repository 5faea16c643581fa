#!/usr/bin/env python3
"""Cost of the fused prologue/epilogue options (GEN I/O path) against plain transforms."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as orc
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec, vector as V
for dtype, n in ((np.float32, 1 << 24), (np.float64, 1 << 22), (np.float32, 1 << 20)):
    x = orc.fill_uniform(2 * n, 1, -10, 10, dtype)
    v = DspVec(x, is_complex=True)
    def t(fn, inv):
        fn(); inv(); bd.lib.bdsp_hip_synchronize(None)
        ts = []
        for _ in range(5):
            bd.lib.bdsp_hip_synchronize(None); t0 = time.perf_counter(); fn(); bd.lib.bdsp_hip_synchronize(None)
            ts.append(time.perf_counter() - t0); inv()
        return sorted(ts)[2] * 1e6
    print("%s n=%d: plain_fft %.1f us, fft(shift) %.1f us, windowed_fft(Hann) %.1f us, ifft(scale+shift) %.1f us" % (
        np.dtype(dtype).name, n,
        t(v.plain_fft, v.plain_ifft), t(v.fft, v.ifft), t(lambda: v.windowed_fft(V.WINDOW_HANN), v.ifft),
        (v.fft(), t(v.ifft, v.fft))[1]))
