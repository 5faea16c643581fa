"""C2 under rocprofv3: per-kernel durations of the two passes of one (or `batch`) 1M-point fft -> magnitude.
usage (GPU box): rocprofv3 --kernel-trace --stats -d out -- python3 tools/c2_prof.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basic_dsp_amd as bd
lib = bd.lib
n = 1 << 20
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
xs = [torch.rand(2 * n * b, device="cuda") * 20 - 10 for _ in range(3)]
y = torch.empty(2 * n * b, device="cuda")
sp = bd._lib.torch_stream_arg()
flag = C.c_int(0)
for i in range(300):
    bd._lib.check(lib.bdsp_hip_dev_fft(0, xs[i % 3].data_ptr(), y.data_ptr(), n, b, bd._lib.FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp))
torch.cuda.synchronize()
