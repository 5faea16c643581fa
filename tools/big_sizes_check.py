import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
from basic_dsp_amd import DspVec
import basic_dsp_amd as bd
for bits in (25, 27, 28):
    n = 1 << bits
    x = orc.fill_uniform(2 * n, 3, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    t0 = time.perf_counter()
    assert v.plain_fft() == 0
    bd.lib.bdsp_hip_synchronize(None)
    t1 = time.perf_counter()
    X = v.datac()
    # check two bins against the definition
    idx = np.arange(n, dtype=np.int64)
    xc = x.view(np.complex64)
    for k in (1, n // 3):
        ref = np.sum(xc.astype(np.complex128) * np.exp(-2j * np.pi * ((idx * k) % n) / n))
        assert abs(X[k] - ref) / (np.sqrt(n) * 10) < 3e-6, (bits, k)
    assert v.plain_ifft() == 0 and v.scale(1.0 / n) == 0
    err = np.linalg.norm(v.data().astype(np.float64) - x) / np.linalg.norm(x)
    print("2^%d: fft %.1f ms (first call), round trip rel-L2 %.2e" % (bits, (t1 - t0) * 1e3, err))
    assert err < 3e-6
    h = orc.fill_uniform(2 * 257, 4, -1, 1, np.float32) / 257
    c = DspVec(x, is_complex=True)
    assert c.convolve_signal(DspVec(h, is_complex=True)) == 0
    y = c.data()
    for first in (0, n - 2000, n // 2 + 12345):
        ref = orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), True, first, 2000)
        e = np.linalg.norm(y[2 * first:2 * (first + 2000)] - ref) / np.linalg.norm(ref)
        assert e < 1e-6, (bits, first, e)
    print("   conv ok")
    del v, c, x, X, y
