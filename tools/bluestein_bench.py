import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
from basic_dsp_amd import DspVec
import basic_dsp_amd as bd
for n in (1000, 1920, 6000, 10000, 30000, 100000, 360000, 1000000, 1000003, 3000000, 3145728, 10000000):
    x = orc.fill_uniform(2 * n, 3, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    v.plain_fft(); v.plain_ifft()
    bd.lib.bdsp_hip_synchronize(None)
    t0 = time.perf_counter()
    for _ in range(10):
        v.plain_fft(); v.plain_ifft()
    bd.lib.bdsp_hip_synchronize(None)
    print("n=%d: %.1f us per transform" % (n, (time.perf_counter() - t0) / 20 * 1e6))
