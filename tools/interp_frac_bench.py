import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
from basic_dsp_amd import DspVec
import basic_dsp_amd as bd
n = 1 << 22
for dtype in (np.float32, np.float64):
    x = orc.fill_uniform(2 * n, 1, -10, 10, dtype)
    for fid, ro, name in ((0, 0.0, "sinc"), (1, 0.35, "rc")):
        vs = [DspVec(x, is_complex=True) for _ in range(4)]
        vs[0].interpolatef(fid, 2.5, 0.0, 12, rolloff=ro)
        bd.lib.bdsp_hip_synchronize(None)
        t0 = time.perf_counter()
        for v in vs[1:]:
            v.interpolatef(fid, 2.5, 0.0, 12, rolloff=ro)
        bd.lib.bdsp_hip_synchronize(None)
        print("%s %s factor 2.5, 4M -> 10M: %.1f us" % (np.dtype(dtype).name, name, (time.perf_counter() - t0) / 3 * 1e6))
