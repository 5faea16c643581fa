import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
from basic_dsp_amd import DspVec
import basic_dsp_amd as bd
vs = []
for k in range(12):
    v = DspVec(is_complex=True, dtype=np.float32, length=1 << 30)   # 4 GiB per buffer, two buffers per handle
    assert v.scale(2.0) == 0
    vs.append(v)
print("allocated", len(vs))
del vs
x = orc.fill_uniform(2 * 4096, 1, -1, 1, np.float32)
for k in range(200):
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0 and v.plain_ifft() == 0
ref = v.data() / 4096
print("ok", float(np.abs(ref - x).max()))
w = DspVec(is_complex=True, dtype=np.float32, length=1 << 30)
assert w.offset(1.0) == 0
print("realloc ok", w.data()[:2])
