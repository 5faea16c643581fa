#!/usr/bin/env python3
"""Times and CHECKS alternative plans of the power-of-two transform on VALID data, cold and cache-resident (round 5).

    BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab.so python tools/plan_probe.py \
        --bits 21,22 --prec f64 [--batch 16] [--flags 2] [--window 4 0.5] --plans default,2048x4:2048x4,256x16:128x32:128x32 [--tag text]

One row per (size, precision, plan): `us` = every call reads a pristine copy of the random input that was written long ago
and is used exactly once, and ping-pongs through a scratch buffer of its own (a COLD input and a cold trade buffer: the
caches were flushed with a 1 GB fill after the copies were made; `--scratch shared` = one scratch for all calls), one event
pair around the loop; `us_hot` = the input was copied into its buffer right before the call (in the caches, as after a
producer kernel), one event pair per call, median minus the cost of an empty pair.  This is the protocol of
tools/bench_configs.py; rounds 1-3 timed loops of in-place transforms that fed every call the previous call's output, i.e.
inf / NaN after a few dozen calls.

`--plans`: `default` = what the library picks, otherwise super-radix x tile width per pass, passes separated by ':'
(BDSP_FFT_PLAN, LAB build only; read per call, so one process serves all plans).  The switches that are read ONCE per process
(BDSP_FFT_LAST_INPLACE, BDSP_FFT_NO_LAST_INPLACE, BDSP_FFT_NO_CHUNKS, BDSP_FFT_CHUNK_MB, BDSP_FFT_NO_WG4, BDSP_FFT_NO_WGBATCH)
and the compile-time ones (another library: BDSP_HIP_LIBRARY) come from the environment; `--tag` names them in the row.
Every plan is checked: a plain complex transform against numpy (first and last vector of the batch), a transform with fused
options against the default plan's result in the same process."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import basic_dsp_amd as bd

ap = argparse.ArgumentParser()
ap.add_argument("--bits", default=None)
ap.add_argument("--points", default=None, help="any lengths instead of --bits (mixed-radix / chirp-z plans): comma separated")
ap.add_argument("--prec", default="f32")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--inverse", type=int, default=0)
ap.add_argument("--window", nargs=2, default=None, metavar=("ID", "ALPHA"))
ap.add_argument("--plans", default="default")
ap.add_argument("--iters", type=int, default=0)
ap.add_argument("--tag", default="")
ap.add_argument("--json", action="store_true")
ap.add_argument("--scratch", choices=("per-call", "shared"), default="per-call",
                help="cold loop: every call gets its own scratch buffer as well (default) or all calls share one")
ap.add_argument("--only", choices=("both", "cold"), default="both", help="cold: the cold loop is the LAST thing the process runs (for a kernel trace)")
a = ap.parse_args()

lib = bd.lib
sp = bd._lib.torch_stream_arg()
dev = torch.device("cuda", 0)
dt = torch.float32 if a.prec == "f32" else torch.float64
elem = 0 if a.prec == "f32" else 1
esz = 4 if elem == 0 else 8
wid, walpha = (int(a.window[0]), float(a.window[1])) if a.window else (-1, 0.0)
flag = C.c_int(0)
ms = C.c_float(0)
MAG, OUT_REAL = bd._lib.FFT_MAGNITUDE, getattr(bd._lib, "FFT_OUT_REAL", 0)


def ev_overhead():
    v = []
    for _ in range(20):
        e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(e0, sp); lib.bdsp_hip_event_record(e1, sp)
        lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms)); v.append(ms.value * 1e3)
        lib.bdsp_hip_event_destroy(e0); lib.bdsp_hip_event_destroy(e1)
    return min(v)


def set_plan(p):
    if p == "default": os.environ.pop("BDSP_FFT_PLAN", None)
    else: os.environ["BDSP_FFT_PLAN"] = p.replace(":", ",")


junk = torch.empty(1 << 28, device=dev, dtype=torch.float32)  # 1 GB: what flushes the L2s and the 256 MB Infinity Cache
EV0 = ev_overhead()

sizes = [(int(b), 1 << int(b)) for b in a.bits.split(",")] if a.bits else [(None, int(v)) for v in a.points.split(",")]
for bits, n in sizes:
    b = a.batch
    label = ("2^%-2d" % bits) if bits is not None else ("%-9d" % n)
    g = torch.Generator(device=dev).manual_seed(5 + (bits if bits is not None else n % 1000))
    pristine = torch.rand(2 * n * b, device=dev, dtype=dt, generator=g) * 20 - 10
    y = torch.empty(2 * n * b, device=dev, dtype=dt)
    vec_bytes = 2 * n * b * esz
    iters = a.iters or max(6, min(30, int(3e9 // vec_bytes)))  # (x 2 buffers per call + the flush)

    def call(buf, scratch=None):
        return lib.bdsp_hip_dev_fft(elem, buf.data_ptr(), (y if scratch is None else scratch).data_ptr(), n, b, a.flags, 1.0, wid, walpha, C.byref(flag), sp)

    def result(buf):
        """what the call left, as a flat array of scalars of the output's size"""
        out = y if flag.value else buf
        per = n if (a.flags & (MAG | OUT_REAL)) else 2 * n
        return out[: per * b].double().cpu().numpy()

    base = None
    for plan in a.plans.split(","):
        set_plan(plan)
        w = pristine.clone()
        rc = call(w); torch.cuda.synchronize()
        row = {"bits": bits, "points": n, "prec": a.prec, "batch": b, "flags": a.flags, "window": wid, "plan": plan, "tag": a.tag}
        if rc != 0:
            row["rc"] = rc
            print(json.dumps(row) if a.json else "%s %s x%-3d %-28s %-22s rc=%d (unsupported)" % (label, a.prec, b, plan, a.tag, rc), flush=True)
            continue
        res = result(w)
        if a.flags == 0 and wid < 0 and not a.inverse:
            src = pristine.double().cpu().numpy().view(np.complex128).reshape(b, n)
            got = res.view(np.complex128).reshape(b, n)
            err = 0.0
            for v in sorted({0, b - 1}):
                ref = np.fft.fft(src[v])
                err = max(err, float(np.linalg.norm(got[v] - ref) / np.linalg.norm(ref)))
            del src, got
        else:
            if base is None: base = res
            err = float(np.linalg.norm(res - base) / np.linalg.norm(base)) if plan != "default" or base is not res else 0.0
        del w
        # warm the clock and the code: 0.15 s of calls on restored buffers
        warm = [pristine.clone() for _ in range(2)]
        import time
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 0.15:
            for _ in range(3):
                warm[k % 2].copy_(pristine); call(warm[k % 2]); k += 1
            torch.cuda.synchronize()
        # cold: every input valid, written long ago, used once
        # ... and so is its scratch buffer (a B2 handle's trade buffer is as cold as its data).  With ONE shared scratch a plan
        # that leaves its result IN the scratch buffer (the in-place last pass) never writes a result back to HBM in this
        # loop -- the next call overwrites it in the Infinity Cache -- while a plan that leaves it in the (rotating) input
        # buffer pays for the write-back of the previous call's result: run 2 of round 5 measured that artefact, not the plan
        bufs = [pristine.clone() for _ in range(iters)]
        scr = [torch.empty_like(y) for _ in range(iters)] if a.scratch == "per-call" else [None] * iters
        for t in scr:
            if t is not None: t.copy_(y)
        junk.fill_(1.0)
        for _ in range(2):  # (the clock again, on buffers that do not matter)
            warm[0].copy_(pristine); call(warm[0])
        junk.fill_(2.0)
        e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(e0, sp)
        for i in range(iters): call(bufs[i], scr[i])
        lib.bdsp_hip_event_record(e1, sp)
        lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        cold = ms.value / iters * 1e3
        del bufs, scr
        if a.only == "cold":
            torch.cuda.synchronize()
            print("%s %s x%-3d fl=%-2d win=%-2d %-28s %-22s cold %8.2f us" % (label, a.prec, b, a.flags, wid, plan, a.tag, cold), flush=True)
            continue
        # hot: the input written right before the call
        pairs = []
        for i in range(iters):
            warm[i % 2].copy_(pristine)
            p0, p1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
            lib.bdsp_hip_event_record(p0, sp); call(warm[i % 2]); lib.bdsp_hip_event_record(p1, sp)
            pairs.append((p0, p1))
        torch.cuda.synchronize()
        d = []
        for p0, p1 in pairs:
            lib.bdsp_hip_event_elapsed_ms(p0, p1, C.byref(ms)); d.append(ms.value * 1e3)
            lib.bdsp_hip_event_destroy(p0); lib.bdsp_hip_event_destroy(p1)
        d.sort()
        hot = d[len(d) // 2] - EV0
        del warm
        row.update({"us": round(cold, 2), "us_hot": round(hot, 2), "rel_l2": err, "iters": iters})
        if a.json: print(json.dumps(row), flush=True)
        else:
            print("%s %s x%-3d fl=%-2d win=%-2d %-28s %-22s cold %8.2f us   hot %8.2f us   err %.2e" %
                  (label, a.prec, b, a.flags, wid, plan, a.tag, cold, hot, err), flush=True)
    del pristine, y
    torch.cuda.empty_cache()
