import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import basic_dsp_amd as bd
lib = bd.lib
lib.bdsp_hip_debug_conv_timeline.argtypes = [C.c_void_p]
n, m = 1 << 24, 1024
dev = torch.device("cuda", 0)
x = torch.rand(2 * n, device=dev) * 20 - 10
y = torch.empty(2 * n, device=dev)
taps = (torch.rand(2 * m, device=dev) * 2 - 1) / m
spec = torch.empty(2 * 4096, device=dev)
sp = bd._lib.torch_stream_arg()
lib.bdsp_hip_dev_conv_prepare(0, taps.data_ptr(), m, spec.data_ptr(), sp)
for _ in range(3):
    lib.bdsp_hip_dev_convolve_prepared(0, x.data_ptr(), y.data_ptr(), n, 1, spec.data_ptr(), m, sp)
dbg = torch.zeros(2 * 4 * 512, device=dev, dtype=torch.int64)
lib.bdsp_hip_debug_conv_timeline(C.c_void_p(dbg.data_ptr()))
lib.bdsp_hip_dev_convolve_prepared(0, x.data_ptr(), y.data_ptr(), n, 1, spec.data_ptr(), m, sp)
torch.cuda.synchronize()
lib.bdsp_hip_debug_conv_timeline(None)
d = dbg.cpu().numpy().reshape(2, 4, 512)
for wg in range(2):
    for w in range(4):
        t = d[wg, w]
        k = (t != 0).sum()
        t = t[:k]
        print("WG %d wave %d: %d stamps, total %.1f us (@100MHz counter?)" % (wg, w, k, (t[-1] - t[0]) / 100.0))
        if w == 0:
            NS = 6
            t = t[: (k // NS) * NS].reshape(-1, NS)
            dt = np.diff(t, axis=1)
            gap = t[1:, 0] - t[:-1, NS - 1]
            print("  phase deltas (ticks) mean over blocks:", dt.mean(axis=0).round(0))
            print("  loop-top gap (store issue -> next block start):", gap.mean().round(0), " per-block total:", (t[1:, 0] - t[:-1, 0]).mean().round(0))
