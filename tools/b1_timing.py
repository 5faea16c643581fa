import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
from basic_dsp_amd import vector as V
n = 1 << 24
x = orc.fill_uniform(2 * n, 3, -10, 10, np.float32)
h = orc.fill_uniform(2 * 1024, 4, -1, 1, np.float32)
V.gpu_fft(x.copy())
for name, fn in (("gpu_fft 16M", lambda: V.gpu_fft(x)), ("gpu_convolve_vector 16M x 1024", lambda: V.gpu_convolve_vector(x, h, True))):
    fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    print("%s: %.2f ms  (%.1f GB/s of host traffic)" % (name, dt * 1e3, 2 * x.nbytes / dt / 1e9))
