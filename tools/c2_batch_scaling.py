import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import basic_dsp_amd as bd
from basic_dsp_amd._lib import FFT_MAGNITUDE
lib = bd.lib; sp = bd._lib.torch_stream_arg(); flag = C.c_int(0)
n = 1 << 20
for b in (1, 2, 3, 4, 8):
    xs = [torch.rand(2 * n * b, device="cuda") * 20 - 10 for _ in range(3)]
    sc = torch.empty(2 * n * b, device="cuda")
    def run(i): lib.bdsp_hip_dev_fft(0, xs[i % 3].data_ptr(), sc.data_ptr(), n, b, FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 0.15:
        for _ in range(10): run(k); k += 1
        torch.cuda.synchronize()
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(50): run(i)
    lib.bdsp_hip_event_record(e1, sp)
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    print("batch %d x 1M fft->magnitude: %.1f us (%.1f per vector)" % (b, ms.value / 50 * 1e3, ms.value / 50 * 1e3 / b))
