#!/usr/bin/env python3
"""Is a vector that was just UPLOADED (overwrite_data: a host-to-device copy) cold for the first pass of its transform?
Through the facade (B2), f64 4M points windowed_fft(Hann) and plain_fft at a few sizes: the upload, a synchronise, then one
event pair on the library's stream around the call; median of 15.  Run once per library (BDSP_HIP_LIBRARY): the LAB build
against the one whose first pass loads non-temporally (libbasic_dsp_hip_lab_ntload2.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec
from basic_dsp_amd import vector as V
lib = bd.lib
ms = C.c_float(0)
rng = np.random.default_rng(2)
for dtype, bits, what in ((np.float64, 22, "windowed_fft(Hann)"), (np.float64, 22, "plain_fft"), (np.float64, 21, "plain_fft"), (np.float32, 22, "plain_fft"),
                          (np.float32, 23, "plain_fft"), (np.float64, 23, "plain_fft")):
    n = 1 << bits
    x = (rng.random(2 * n) * 20 - 10).astype(dtype)
    v = DspVec(x, is_complex=True)
    d = []
    for it in range(18):
        v2 = DspVec(x, is_complex=True)  # a fresh handle: new + upload, both buffers untouched by any kernel
        lib.bdsp_hip_synchronize(None)
        e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(e0, None)
        rc = v2.windowed_fft(V.WINDOW_HANN) if what.startswith("windowed") else v2.plain_fft()
        lib.bdsp_hip_event_record(e1, None)
        lib.bdsp_hip_synchronize(None)
        assert rc == 0
        lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        if it >= 3: d.append(ms.value * 1e3)
        lib.bdsp_hip_event_destroy(e0); lib.bdsp_hip_event_destroy(e1)
        del v2
    d.sort()
    print("%-28s %s 2^%d %-20s first call after the upload: median %7.1f us  (min %.1f, max %.1f)" %
          (os.path.basename(bd.LIB_PATH), np.dtype(dtype).name, bits, what, d[len(d) // 2], d[0], d[-1]), flush=True)
