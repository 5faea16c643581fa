import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
from basic_dsp_amd import DspVec
import basic_dsp_amd as bd
for n in (6000, 100000, 1000000):
    x = orc.fill_uniform(2 * n, 3, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    for _ in range(5):
        v.plain_fft(); v.plain_ifft()
    bd.lib.bdsp_hip_synchronize(None)
