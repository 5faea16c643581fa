import numpy as np, sys
sys.path.insert(0, "/root/repo")
from basic_dsp_amd import DspVec, vector as V
x = DspVec(np.random.rand(2 << 20).astype(np.float32), is_complex=True)
h = DspVec(np.random.rand(2 * 1024).astype(np.float32) / 1024, is_complex=True)
assert x.convolve_signal(h) == 0
assert x.windowed_fft(V.WINDOW_HANN) == 0
stats = x.statistics()
spectrum = x.data()
print("readme example ok", stats["count"], spectrum.shape)
