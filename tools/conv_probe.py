#!/usr/bin/env python3
"""Round 5: the block kernel's scheduling experiments on the PRODUCT kernel (LAB library), timing + a bit-for-bit check
against the default schedule in the same process.
    BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab.so python tools/conv_probe.py f64|c5|real
  f64   16M complex f64 points (*) 1024 taps (256 MB in, 256 MB out): streamed stores (BDSP_CONV_NTS), a start delay for the
        second dispatch group (BDSP_CONV_STAGGER_US), equal shares, and combinations -- all read per call
  c5    64 x 1M complex f32 (512 MB in / out): streamed stores, the kernel alone and convolve -> fft
  real  16M real f32 samples (*) 1024 real taps through the facade (events on the library's stream); BDSP_CONV_GROUPS and
        BDSP_CONV_REAL_PREP are read once per process: run once per setting, the row carries a checksum to compare"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import basic_dsp_amd as bd
lib = bd.lib
dev = torch.device("cuda", 0)
what = sys.argv[1] if len(sys.argv) > 1 else "f64"
m = 1024
ms = C.c_float(0)


def timed(fn, iters, sp, warm_s=0.3):
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < warm_s:
        for _ in range(5): fn(k); k += 1
        lib.bdsp_hip_synchronize(sp)
    e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
    lib.bdsp_hip_event_record(e0, sp)
    for i in range(iters): fn(i)
    lib.bdsp_hip_event_record(e1, sp)
    lib.bdsp_hip_synchronize(sp)
    lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
    lib.bdsp_hip_event_destroy(e0); lib.bdsp_hip_event_destroy(e1)
    return ms.value / iters * 1e3


def setenv(d):
    for k in ("BDSP_CONV_NTS", "BDSP_CONV_STAGGER_US", "BDSP_CONV_V3"):
        os.environ.pop(k, None)
    os.environ.update(d)


if what == "f64ab":
    # streamed stores against plain ones, INTERLEAVED (a single pair of timings moves by +-5 % on one box): 12 alternations
    sp = bd._lib.torch_stream_arg()
    n, elem = 1 << 24, 1
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.rand(2 * n, device=dev, dtype=torch.float64, generator=g) * 20 - 10 for _ in range(2)]
    taps = (torch.rand(2 * m, device=dev, dtype=torch.float64, generator=g) * 2 - 1) / m
    y = torch.empty(2 * n, device=dev, dtype=torch.float64)
    conv = lambda i: lib.bdsp_hip_dev_convolve(elem, xs[i % 2].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, sp)
    res = {"0": [], "1": []}
    for rep in range(12):
        for mode in ("0", "1") if rep % 2 == 0 else ("1", "0"):
            os.environ["BDSP_CONV_NTS"] = mode
            res[mode].append(timed(conv, 15, sp, warm_s=0.15 if rep else 0.4))
    for mode, name in (("0", "plain stores"), ("1", "streamed stores")):
        v = sorted(res[mode])
        print("f64 16M (*) 1024 taps, %-16s median of 12 interleaved runs %6.1f us  (min %.1f, max %.1f)  all: %s" %
              (name, (v[5] + v[6]) / 2, v[0], v[-1], " ".join("%.1f" % t for t in res[mode])), flush=True)
    sys.exit(0)

if what in ("f64", "c5"):
    sp = bd._lib.torch_stream_arg()
    if what == "f64": n, b, dt, elem, nbuf, iters = 1 << 24, 1, torch.float64, 1, 2, 15
    else: n, b, dt, elem, nbuf, iters = 1 << 20, 64, torch.float32, 0, 2, 10
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.rand(2 * n * b, device=dev, dtype=dt, generator=g) * 20 - 10 for _ in range(nbuf)]
    taps = (torch.rand(2 * m, device=dev, dtype=dt, generator=g) * 2 - 1) / m
    y = torch.empty(2 * n * b, device=dev, dtype=dt)
    scr = torch.empty(2 * n * b, device=dev, dtype=dt) if what == "c5" else None
    flag = C.c_int(0)
    ref = None
    variants = [("default", {}, None), ("streamed stores", {"BDSP_CONV_NTS": "1"}, None)]
    if what == "f64":
        for us in (2, 4, 6, 8, 12):
            variants.append(("second group starts %d us later" % us, {"BDSP_CONV_STAGGER_US": str(us)}, None))
        variants += [("equal shares", {}, (50, 0)), ("equal shares, second group 6 us later", {"BDSP_CONV_STAGGER_US": "6"}, (50, 0)),
                     ("shares 60 / 40", {}, (60, 0)), ("shares 52 / 48", {}, (52, 0)),
                     ("streamed stores, second group 6 us later", {"BDSP_CONV_NTS": "1", "BDSP_CONV_STAGGER_US": "6"}, None),
                     ("streamed stores, equal shares", {"BDSP_CONV_NTS": "1"}, (50, 0)),
                     ("512 threads x 8 points (k_overlap_save_v3)", {"BDSP_CONV_V3": "1"}, None),
                     ("512 threads x 8 points, streamed stores", {"BDSP_CONV_V3": "1", "BDSP_CONV_NTS": "1"}, None),
                     ("512 threads x 8 points, equal shares", {"BDSP_CONV_V3": "1"}, (50, 0)),
                     ("default (again)", {}, None)]
    for name, env, shares in variants:
        setenv(env)
        def conv(i):
            if shares: return lib.bdsp_hip_dev_convolve_ex(elem, xs[i % nbuf].data_ptr(), y.data_ptr(), n, b, taps.data_ptr(), m, shares[0], shares[1], sp)
            return lib.bdsp_hip_dev_convolve(elem, xs[i % nbuf].data_ptr(), y.data_ptr(), n, b, taps.data_ptr(), m, sp)
        assert conv(0) == 0, bd._lib.last_error()
        torch.cuda.synchronize()
        if ref is None: ref = y.clone(); same = True
        else: same = bool(torch.equal(y, ref))
        err = 0.0 if same else float(torch.linalg.vector_norm((y - ref).double()) / torch.linalg.vector_norm(ref.double()))
        us = timed(conv, iters, sp)
        line = "%-4s %-46s %8.1f us  (%.3f of 8 TB/s)  result %s" % (what, name, us, 2 * n * b * (16 if elem else 8) / us / 1e3 / 8000, "bit-identical" if same else "rel-L2 %.1e from the default's" % err)
        if what == "c5":
            def both(i):
                conv(i)
                lib.bdsp_hip_dev_fft(elem, y.data_ptr(), scr.data_ptr(), n, b, 0, 1.0, -1, 0.0, C.byref(flag), sp)
            line += "   convolve -> fft %8.1f us" % timed(both, iters, sp)
        print(line, flush=True)
    if what == "f64":  # the headline pair with its 128 MB result streamed (round 3: +8 us), for the record
        del xs, y, ref
        n = 1 << 24
        xs = [torch.rand(2 * n, device=dev, dtype=torch.float32) * 20 - 10 for _ in range(3)]
        y = torch.empty(2 * n, device=dev, dtype=torch.float32); scr = torch.empty_like(y)
        t32 = taps.float()
        for name, env in (("default", {}), ("streamed stores", {"BDSP_CONV_NTS": "1"})):
            setenv(env)
            def both(i):
                lib.bdsp_hip_dev_convolve(0, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, t32.data_ptr(), m, sp)
                lib.bdsp_hip_dev_fft(0, y.data_ptr(), scr.data_ptr(), n, 1, 0, 1.0, -1, 0.0, C.byref(flag), sp)
            print("f32 16M headline step, %-30s %8.1f us" % (name, timed(both, 40, sp)), flush=True)
else:
    from basic_dsp_amd import DspVec
    n = 1 << 24
    rng = np.random.default_rng(4)
    x = (rng.random(n, dtype=np.float32) * 20 - 10)
    h = ((rng.random(m, dtype=np.float32) * 2 - 1) / m).astype(np.float32)
    src = [DspVec(x) for _ in range(3)]
    work = [DspVec(x) for _ in range(3)]
    hv = DspVec(h)
    w = src[0].clone(); assert w.convolve_signal(hv) == 0
    out = w.data()
    chk = float(np.sum(np.abs(out.astype(np.float64)))), float(out[12345]), float(out[-1])
    del w
    # every call convolves a vector that holds the random input (the facade works in place, so each call gets a copy made
    # by the device-to-device clone the facade offers; the copies are made before the timed loop)
    iters = 60
    def run_loop(k):
        ws = [src[i % 3].clone() for i in range(k)]
        lib.bdsp_hip_synchronize(None)
        e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(e0, None)
        for v in ws: v.convolve_signal(hv)
        lib.bdsp_hip_event_record(e1, None)
        lib.bdsp_hip_synchronize(None)
        lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        return ms.value / k * 1e3
    for _ in range(4): run_loop(30)
    us = sorted(run_loop(iters) for _ in range(5))
    t0 = time.perf_counter()
    ws = [src[i % 3].clone() for i in range(iters)]
    lib.bdsp_hip_synchronize(None)
    t0 = time.perf_counter()
    for v in ws: v.convolve_signal(hv); lib.bdsp_hip_synchronize(None)
    host = (time.perf_counter() - t0) / iters * 1e6
    print("real f32 16M (*) %d real taps  groups=%s prep=%s   kernel(s) %6.1f us median of 5 loops (min %.1f, max %.1f; events on the library stream)   call + synchronise %6.1f us   checksum %.6e %.6e %.6e" %
          (m, os.environ.get("BDSP_CONV_GROUPS", "3"), os.environ.get("BDSP_CONV_REAL_PREP", "0"), us[2], us[0], us[-1], host, *chk), flush=True)
