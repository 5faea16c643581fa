"""Wall time of the B1 (host-slice) entry points on a 16M-point f32 vector: what a GpuSupport<T> caller sees."""
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
import basic_dsp_amd as bd
from basic_dsp_amd import vector as V
n = 1 << 24
x = orc.fill_uniform(2 * n, 3, -10, 10, np.float32)
h = orc.fill_uniform(2 * 1024, 4, -1, 1, np.float32) / 1024
y = np.ones_like(x)  # an existing, touched target like a Rust caller's buffer
rs, re = C.c_size_t(0), C.c_size_t(0)
P = lambda a: a.ctypes.data_as(C.c_void_p)
def conv():
    assert bd.lib.bdsp_hip_convolve_vector_f32(1, P(x), x.size, P(y), y.size, P(h), h.size, C.byref(rs), C.byref(re)) == 1
V.gpu_fft(x.copy())
for name, fn in (("gpu_fft 16M", lambda: V.gpu_fft(x)), ("gpu_convolve_vector 16M x 1024", conv)):
    fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    print("%s: %.2f ms  (%.1f GB/s of host traffic)" % (name, dt * 1e3, 2 * x.nbytes / dt / 1e9))
# the pipelined result equals the device-resident path
x = orc.fill_uniform(2 * n, 3, -10, 10, np.float32)
conv()
v = V.DspVec(x, is_complex=True); assert v.convolve_signal(V.DspVec(h, is_complex=True)) == 0
print("max |B1 - B2| =", float(np.max(np.abs(v.data() - y))))
