#!/usr/bin/env python3
"""What a fresh process's FIRST calls cost, split into code-object load and allocation (round 6).

Each scenario runs in a fresh child process (numpy + ctypes, no torch): device init + a first tiny upload, then a sequence of
first calls, each followed by a stream synchronise.  HIP loads a code object when the first kernel OUT OF IT is launched
(one code object per translation unit of the library: fft_f32, fft_f64, conv_v2, elementwise, ...), so the order of the calls
tells the load of one object from the load of another, and a SMALL vector tells the load from the 128 MB workspace blocks a
16M-point call allocates.   python3 tools/first_call_probe.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec
seq = sys.argv[2].split(",")
rng = np.random.default_rng(1)
def vec(n, cplx=True, dt=np.float32): return DspVec((rng.random((2 if cplx else 1) * n) * 2 - 1).astype(dt), is_complex=cplx)
t = time.perf_counter(); bd.require_gpu(); tiny = vec(16); bd.lib.bdsp_hip_synchronize(None)
out = [("device_init+tiny_upload", (time.perf_counter() - t) * 1e3)]
big, bigc, small, small64 = None, None, None, None
h = vec(1024)
for name in seq:
    if name.endswith("16m") and big is None: big = vec(1 << 24); bigc = vec(1 << 24)
    if name.endswith("4k") and small is None: small = vec(4096); small64 = vec(4096, dt=np.float64)
    bd.lib.bdsp_hip_synchronize(None)
    t = time.perf_counter()
    if name == "scale_4k": assert small.scale(2.0) == 0
    elif name == "swap_4k": assert small.swap_halves() == 0
    elif name == "fft_4k": assert small.plain_fft() == 0
    elif name == "fft64_4k": assert small64.plain_fft() == 0
    elif name == "ifft_4k": assert small.plain_ifft() == 0
    elif name == "conv_4k": assert vec(1 << 14).convolve_signal(h) == 0
    elif name == "fft_16m": assert big.plain_fft() == 0
    elif name == "ifft_16m": assert big.plain_ifft() == 0
    elif name == "conv_16m": assert bigc.convolve_signal(h) == 0
    else: raise SystemExit("unknown step " + name)
    bd.lib.bdsp_hip_synchronize(None)
    out.append((name, (time.perf_counter() - t) * 1e3))
print("PROBE " + json.dumps(out))
"""
SCENARIOS = [
    "scale_4k,fft_4k,ifft_4k,fft_16m,ifft_16m,conv_4k,conv_16m",
    "fft_16m,ifft_16m,fft_16m,conv_16m,conv_16m",
    "conv_16m,fft_16m,ifft_16m",
    "fft64_4k,fft_4k,swap_4k,conv_4k",
]
for sc in SCENARIOS:
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, sc], capture_output=True, text=True, timeout=300)
    rows = [json.loads(l[6:]) for l in p.stdout.splitlines() if l.startswith("PROBE ")]
    if not rows:
        print("FAILED", sc, p.stderr[-500:]); continue
    print("  ".join("%s %.2f ms" % (n, ms) for n, ms in rows[0]))
