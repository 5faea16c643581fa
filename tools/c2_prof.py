import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, ctypes as C
import basic_dsp_amd as bd
lib = bd.lib
n = 1 << 20
dev = torch.device("cuda", 0)
x = torch.rand(2 * n, device=dev) * 20 - 10
y = torch.empty(2 * n, device=dev)
z = torch.empty(n, device=dev)
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
for _ in range(10):
    bd._lib.check(lib.bdsp_hip_dev_fft(0, x.data_ptr(), y.data_ptr(), n, 1, bd._lib.FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp))
torch.cuda.synchronize()
