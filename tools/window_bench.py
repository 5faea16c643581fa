#!/usr/bin/env python3
"""GPU box: cost of windowed_fft with each reference window against plain_fft (facade calls, median of 7, host clock)."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as orc, basic_dsp_amd as bd
from basic_dsp_amd import DspVec, vector as V
for dtype, n in ((np.float32, 1 << 24), (np.float64, 1 << 22)):
    x = orc.fill_uniform(2 * n, 1, -10, 10, dtype)
    v = DspVec(x, is_complex=True)
    def t(fn, inv):
        fn(); inv(); bd.lib.bdsp_hip_synchronize(None); ts = []
        for _ in range(7):
            bd.lib.bdsp_hip_synchronize(None); t0 = time.perf_counter(); fn(); bd.lib.bdsp_hip_synchronize(None); ts.append(time.perf_counter() - t0); inv()
        return sorted(ts)[3] * 1e6
    print("%s n=%d (%s): plain_fft %.1f us; windowed_fft triangular %.1f, Hamming %.1f, Blackman-Harris %.1f, Hann %.1f us" % (
        np.dtype(dtype).name, n, os.path.basename(bd.LIB_PATH), t(v.plain_fft, v.plain_ifft), t(lambda: v.windowed_fft(0), v.ifft),
        t(lambda: v.windowed_fft(1), v.ifft), t(lambda: v.windowed_fft(2), v.ifft), t(lambda: v.windowed_fft(4), v.ifft)))
    v.fft()
    print("      inverses: plain_ifft %.1f us, ifft %.1f; windowed_ifft triangular %.1f, Hamming %.1f, Blackman-Harris %.1f, Hann %.1f us" % (
        t(v.plain_ifft, v.plain_fft), t(v.ifft, v.fft), t(lambda: v.windowed_ifft(0), v.fft), t(lambda: v.windowed_ifft(1), v.fft),
        t(lambda: v.windowed_ifft(2), v.fft), t(lambda: v.windowed_ifft(4), v.fft)))
