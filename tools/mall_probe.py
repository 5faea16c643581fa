"""Does a working set that fits the 256 MB Infinity Cache stream faster than HBM?  In-place scale on one buffer."""
import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
sp = bd._lib.torch_stream_arg()
for mb in (8, 16, 32, 64, 128, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.rand(n, device=dev)
    f = lambda: lib.bdsp_hip_dev_real_scale(0, x.data_ptr(), n, 1.0000001, sp)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(20): f()
        torch.cuda.synchronize()
    iters = max(20, int(2e9 // (mb << 20)))
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for _ in range(iters): f()
    lib.bdsp_hip_event_record(e1, sp)
    torch.cuda.synchronize()
    ms = C.c_float(0); lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    us = ms.value / iters * 1e3
    print("%5d MB in place: %8.1f us  %6.0f GB/s (read + write)" % (mb, us, 2.0 * mb * (1 << 20) / us / 1e3))
    del x
