#!/usr/bin/env python3
"""Where does the B1 boundary (GpuSupport<T> on host slices) beat the CPU path the reference would run instead?

The reference sends every complex vector of >= 10 000 scalars to T::fft (vector/src/vector_types/time_freq/mod.rs:41-44)
and every convolution with len > 10 000 to T::gpu_convolve_vector (convolution.rs:504-527) once a backend reports
has_gpu_support.  A B1 call is a host round trip -- upload, kernels, download, one synchronisation -- so below some length
the caller's own CPU code wins.  This tool measures, on the box it runs on:

  fft   wall time of bdsp_hip_fft_{f32,f64}, forward and inverse, 2^12 ... 2^24 points + 5 000 / 10 000 / 100 003
  conv  wall time of bdsp_hip_convolve_vector_{f32,f64}, N x M complex and real
  cpu   the same calls on ONE host core through the oracle (kind "port": orc_fft is a radix-2 / Bluestein restatement,
        rustfft itself is not in this image; orc_convolve_signal restates convolution.rs:464-543 with has_gpu = 0, i.e.
        exactly what runs when the plug-in declines) and through numpy / scipy (pocketfft: "sanity", NOT the reference --
        it stands in for a tuned CPU FFT such as rustfft, so the policy is set against the FASTER of the two CPU rows)

for every staging mode of the LAB library (BDSP_B1_STAGE = 0 pageable copies, 1 pinned copies, 2 kernels read the pinned
stage, 3 kernels read and write it) and for the product library as shipped.  The parent process never touches the GPU; each
variant runs in a child.  Output: a table on stdout and JSON lines in --out.

    python3 tools/b1_crossover.py --out gpurun_out/r06_b1/crossover.jsonl > gpurun_out/r06_b1/crossover.txt
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FFT_POINTS = [1 << k for k in range(12, 25)] + [5000, 10000, 100003]
CONV_N_COMPLEX = [5001, 1 << 14, 1 << 17, 1 << 20]   # 5 001 points = 10 002 scalars: the first length convolve_signal dispatches
CONV_N_REAL = [10001, 1 << 14, 1 << 17, 1 << 20]
CONV_M = [3, 5, 128, 1024, 4096]


def timed(fn, reset, budget_s=0.12, min_reps=5, max_reps=200):
    """min / median wall time of fn(); reset() runs untimed before every call."""
    for _ in range(2):
        reset(); fn()
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < min_reps or (time.perf_counter() < t_end and len(ts) < max_reps):
        reset()
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"min_us": ts[0] * 1e6, "med_us": ts[len(ts) // 2] * 1e6, "reps": len(ts)}


def child_gpu(what, label, args):
    import ctypes as C
    import numpy as np
    import oracle_lib as orc
    import basic_dsp_amd as bd
    bd.require_gpu()
    for key in range(4):
        bd.lib.bdsp_hip_b1_policy_set(key, 0)  # measure every size: the policy is what this run is FOR
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rows = []
    for dtype, sfx in ((np.float32, "f32"), (np.float64, "f64")):
        if what == "fft":
            fft = getattr(bd.lib, "bdsp_hip_fft_" + sfx)
            for n in FFT_POINTS:
                if args.quick and n > (1 << 20):
                    continue
                x0 = orc.fill_uniform(2 * n, 7 + n % 97, -1, 1, dtype)
                x = x0.copy()
                for inverse in (0, 1):
                    r = timed(lambda: fft(1, P(x), x.size, inverse), lambda: np.copyto(x, x0))
                    rows.append(dict(r, op="fft", dtype=sfx, points=n, inverse=inverse, variant=label))
        else:
            conv = getattr(bd.lib, "bdsp_hip_convolve_vector_" + sfx)
            rs, re = C.c_size_t(0), C.c_size_t(0)
            for is_complex, ns in ((1, CONV_N_COMPLEX), (0, CONV_N_REAL)):
                e = 2 if is_complex else 1
                for n in ns:
                    if args.quick and n > (1 << 17):
                        continue
                    x = orc.fill_uniform(e * n, 11 + n % 89, -1, 1, dtype)
                    y = np.ones_like(x)
                    for m in CONV_M:
                        if m > n:
                            continue
                        h = orc.fill_uniform(e * m, 13 + m, -1, 1, dtype) / dtype(m)
                        code = conv(is_complex, P(x), x.size, P(y), y.size, P(h), h.size, C.byref(rs), C.byref(re))
                        assert code == 1, (code, bd.last_error())
                        r = timed(lambda: conv(is_complex, P(x), x.size, P(y), y.size, P(h), h.size, C.byref(rs), C.byref(re)),
                                  lambda: None)
                        rows.append(dict(r, op="conv", dtype=sfx, points=n, taps=m, is_complex=is_complex, variant=label))
    for r in rows:
        print("ROW " + json.dumps(r), flush=True)


def child_cpu(what, args):
    """One host core: the oracle (port) and numpy / scipy (sanity).  No GPU, no product library."""
    import ctypes as C
    import numpy as np
    import oracle_lib as orc
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rows = []
    for dtype, sfx in ((np.float32, "f32"), (np.float64, "f64")):
        cdt = np.complex64 if dtype == np.float32 else np.complex128
        if what == "fft":
            offt = orc._fn("orc_fft", dtype)
            offt.argtypes = [C.c_void_p, C.c_size_t, C.c_int]; offt.restype = None
            for n in FFT_POINTS:
                if args.quick and n > (1 << 20):
                    continue
                x0 = orc.fill_uniform(2 * n, 7 + n % 97, -1, 1, dtype)
                x = x0.copy()
                xc = x0.view(cdt)
                out = np.empty_like(xc)
                big = n >= (1 << 22)
                for inverse in (0, 1):
                    r = timed(lambda: offt(P(x), n, inverse), lambda: np.copyto(x, x0), budget_s=0.1, min_reps=3 if big else 5)
                    rows.append(dict(r, op="fft", dtype=sfx, points=n, inverse=inverse, variant="cpu_port_1core"))
                    f = np.fft.ifft if inverse else np.fft.fft
                    kw = {"norm": "forward"} if inverse else {}  # unnormalised in both directions, like the reference
                    r = timed(lambda: f(xc, out=out, **kw), lambda: None, budget_s=0.1, min_reps=3 if big else 5)
                    rows.append(dict(r, op="fft", dtype=sfx, points=n, inverse=inverse, variant="numpy_sanity"))
        else:
            import scipy.signal as ss
            oconv = orc._fn("orc_convolve_signal", dtype)
            oconv.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
            oconv.restype = C.c_int
            for is_complex, ns in ((1, CONV_N_COMPLEX), (0, CONV_N_REAL)):
                e = 2 if is_complex else 1
                for n in ns:
                    if args.quick and n > (1 << 17):
                        continue
                    x = orc.fill_uniform(e * n, 11 + n % 89, -1, 1, dtype)
                    y = np.zeros_like(x)
                    for m in CONV_M:
                        if m > n:
                            continue
                        h = orc.fill_uniform(e * m, 13 + m, -1, 1, dtype) / dtype(m)
                        path = C.c_int(0)
                        work = n * m * (4 if is_complex else 1)
                        reps = 1 if work > 2e9 else (3 if work > 2e8 else 5)
                        r = timed(lambda: oconv(P(x), x.size, P(h), h.size, is_complex, P(y), C.byref(path)), lambda: None,
                                  budget_s=0.1, min_reps=reps, max_reps=max(reps, 50)) if reps > 1 else None
                        if r is None:  # one call, no warm-up: seconds of scalar work
                            t0 = time.perf_counter()
                            oconv(P(x), x.size, P(h), h.size, is_complex, P(y), C.byref(path))
                            dt = (time.perf_counter() - t0) * 1e6
                            r = {"min_us": dt, "med_us": dt, "reps": 1}
                        rows.append(dict(r, op="conv", dtype=sfx, points=n, taps=m, is_complex=is_complex,
                                         variant="cpu_port_1core", ref_path={1: "simd", 3: "overlap_discard", 4: "scalar"}[path.value]))
                        xs = x.view(cdt) if is_complex else x
                        hs = h.view(cdt) if is_complex else h
                        f = (lambda: np.convolve(xs, hs, "same")) if m < 16 else (lambda: ss.oaconvolve(xs, hs, "same"))
                        r = timed(f, lambda: None, budget_s=0.1, min_reps=3)
                        rows.append(dict(r, op="conv", dtype=sfx, points=n, taps=m, is_complex=is_complex, variant="numpy_sanity"))
    for r in rows:
        print("ROW " + json.dumps(r), flush=True)


def run_child(what, label, env_extra, args):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, os.path.abspath(__file__), "--child", what, "--label", label] + (["--quick"] if args.quick else [])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.child_timeout)
    rows = [json.loads(l[4:]) for l in p.stdout.splitlines() if l.startswith("ROW ")]
    if p.returncode != 0:
        sys.stderr.write("child %s/%s failed rc=%d\n%s\n" % (what, label, p.returncode, p.stderr[-2000:]))
    return rows


def crossover(points_sorted, gpu, cpu):
    """Smallest measured size from which the GPU wins at every larger measured size (None if it never does)."""
    best = None
    for n in reversed(points_sorted):
        if n in gpu and n in cpu and gpu[n] <= cpu[n]:
            best = n
        else:
            break
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--label", default="product")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--child-timeout", type=int, default=900)
    ap.add_argument("--what", default="fft,conv")
    ap.add_argument("--no-lab", action="store_true")
    args = ap.parse_args()
    if args.child:
        if args.label.startswith("cpu"):
            child_cpu(args.child, args)
        else:
            child_gpu(args.child, args.label, args)
        return
    lab = os.path.join(ROOT, "basic_dsp_amd", "lib", "libbasic_dsp_hip_lab.so")
    variants = [("product", {})]
    if os.path.exists(lab) and not args.no_lab:
        variants += [("lab_stage%d" % k, {"BDSP_HIP_LIBRARY": lab, "BDSP_B1_STAGE": str(k)}) for k in (0, 1, 2, 3)]
    rows = []
    for what in args.what.split(","):
        rows += run_child(what, "cpu", {"OMP_NUM_THREADS": "1"}, args)
        for label, env in variants:
            rows += run_child(what, label, env, args)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
    labels = ["cpu_port_1core", "numpy_sanity"] + [v[0] for v in variants]
    print("# B1 crossover, wall time per call in microseconds (min of the repetitions); host cores used by the CPU rows: 1")
    print("# cpu_port_1core = the oracle restatement (kind: port); numpy_sanity = pocketfft / scipy, NOT the reference")
    for sfx in ("f32", "f64"):
        for inverse in (0, 1):
            sel = [r for r in rows if r["op"] == "fft" and r["dtype"] == sfx and r["inverse"] == inverse]
            if not sel:
                continue
            print("\n## fft %s %s" % (sfx, "inverse" if inverse else "forward"))
            print("%10s " % "points" + " ".join("%15s" % l for l in labels))
            tab = {}
            for r in sel:
                tab.setdefault(r["variant"], {})[r["points"]] = r["min_us"]
            pts = sorted({r["points"] for r in sel})
            for n in pts:
                print("%10d " % n + " ".join("%15.1f" % tab.get(l, {}).get(n, float("nan")) for l in labels))
            cpu_fast = {n: min(tab.get("cpu_port_1core", {}).get(n, 1e30), tab.get("numpy_sanity", {}).get(n, 1e30)) for n in pts}
            p2 = [n for n in pts if n & (n - 1) == 0]
            for l in labels[2:]:
                print("crossover %-12s vs port: %s points, vs faster CPU row: %s points (powers of two)" % (
                    l, crossover(p2, tab.get(l, {}), tab.get("cpu_port_1core", {})), crossover(p2, tab.get(l, {}), cpu_fast)))
    for sfx in ("f32", "f64"):
        for is_complex in (1, 0):
            sel = [r for r in rows if r["op"] == "conv" and r["dtype"] == sfx and r["is_complex"] == is_complex]
            if not sel:
                continue
            print("\n## convolve_vector %s %s" % (sfx, "complex" if is_complex else "real"))
            print("%9s %6s %16s " % ("points", "taps", "reference runs") + " ".join("%15s" % l for l in labels))
            keys = sorted({(r["points"], r["taps"]) for r in sel})
            for n, m in keys:
                tab = {r["variant"]: r for r in sel if r["points"] == n and r["taps"] == m}
                path = tab.get("cpu_port_1core", {}).get("ref_path", "?")
                print("%9d %6d %16s " % (n, m, path) + " ".join("%15.1f" % tab.get(l, {}).get("min_us", float("nan")) for l in labels))


if __name__ == "__main__":
    main()
