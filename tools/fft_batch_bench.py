"""Batches of small non-power-of-two transforms through the matrix API (mixed radix vs Bluestein via env)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
from basic_dsp_amd import DspMat
import basic_dsp_amd as bd
for rows, n in ((16384, 1000), (4096, 3000), (65536, 100), (1024, 10000), (2048, 1024)):
    x = orc.fill_uniform(2 * n * rows, 5, -10, 10, np.float32).reshape(rows, 2 * n)
    m = DspMat(x, is_complex=True)
    assert m.plain_fft() == 0 and m.plain_ifft() == 0
    bd.lib.bdsp_hip_synchronize(None)
    t0 = time.perf_counter()
    for _ in range(5):
        assert m.plain_fft() == 0 and m.plain_ifft() == 0
    bd.lib.bdsp_hip_synchronize(None)
    us = (time.perf_counter() - t0) / 10 * 1e6
    print("%6d x n=%5d: %8.1f us per batch  %6.0f GB/s algorithmic" % (rows, n, us, 16.0 * rows * n / us / 1e3))
