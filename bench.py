#!/usr/bin/env python3
"""bench.py -- headline benchmark of basic_dsp_amd (contract: see the build prompt / DESIGN.md 6).

Metric (BASELINE.json): Msamples/s for f32 complex FFT + overlap-save convolution at 16 777 216
points.  One "step" = one pass of the hot path over one synthetic vector per GPU:
    convolve_signal(x, 1024 complex taps)  ->  plain_fft        (config C3 + the 16M-point FFT)
`value` = complex samples that went through the whole step, summed over all ranks, per second,
with every buffer resident in HBM before the timed region.  Inputs rotate through several distinct
buffers so each step reads its input from HBM, not from the 256 MiB Infinity Cache.

N > 1: one process per GPU (torchrun), independent vectors per rank, no data-path collective
(the path shards by vector) -> weak scaling; only the timing barrier uses RCCL.

Extra objects on the JSON line:
  roofline      the dominant kernel (the fused overlap-save launch): algorithmic bytes per launch
                (16 B per complex f32 sample) / mean launch duration measured with HIP
                events on the launch stream inside the timed region, against the 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (a port of the reference algorithm; the Rust reference itself cannot
                be built in this image) timed on a bounded sample of the same workload, rank 0, N=1.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
POINTS = 1 << 24
TAPS = 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=float, default=0.15, help="seconds of untimed load before the warm-up steps (clock ramp)")
    ap.add_argument("--points", type=int, default=POINTS)
    ap.add_argument("--taps", type=int, default=TAPS)
    ap.add_argument("--buffers", type=int, default=3, help="distinct input vectors rotated per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-points", type=int, default=1 << 23)
    return ap.parse_args()


def cpu_baseline(points, taps, sample_points):
    """Oracle (reference algorithm, scalar C, 1 thread = the reference's default
    MultiCoreSettings, threading.rs:210-217) on a bounded sample: overlap_discard with the
    reference's own schedule (scalar head + O(N*M/2) scalar tail + blocks) followed by the FFT."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    n = min(points, sample_points)
    x = orc.fill_uniform(2 * n, 201601171, -10, 10, np.float32)
    h = orc.fill_uniform(2 * taps, 201601172, -1, 1, np.float32) / np.float32(taps)
    t0 = time.perf_counter()
    code, y = orc.overlap_discard(x, h, orc.next_power_of_two(taps), fair=False)
    t1 = time.perf_counter()
    orc.fft(y)
    t2 = time.perf_counter()
    code2, y2 = orc.overlap_discard(x, h, orc.next_power_of_two(taps), fair=True)
    t3 = time.perf_counter()
    assert code == 0 and code2 == 0
    total = t2 - t0
    return {
        "value": n / total / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": "%d-point prefix of the workload: reference-schedule overlap_discard (incl. its "
                  "scalar tail) %.2fs + FFT %.2fs; overlap-save without the scalar tail takes "
                  "%.2fs (%.2f Msamples/s with FFT)" % (n, t1 - t0, t2 - t1, t3 - t2,
                                                        n / ((t3 - t2) + (t2 - t1)) / 1e6),
    }


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel as measured by rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, gfx950 read-size correction applied) -- collected with
    tools/pmc.sh on this same command and committed as profiles/r01_hbm_traffic.json; the live run
    cannot collect counters itself.  None if the profile is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            k = json.load(f)["kernels"]
        for name, v in k.items():
            if name.startswith(kernel_prefix):
                return v["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import basic_dsp_amd as bd
    lib = bd.lib
    bd._lib.check(lib.bdsp_hip_set_device(local_rank), "set_device")
    bd.require_gpu()

    n, m = args.points, args.taps
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(201601171 + rank)
    # synthetic inputs, uniform(-10, 10) like tests/tools/mod.rs:124-139; taps uniform(-1,1)/M
    xs = [(torch.rand(2 * n, generator=gen, device=dev, dtype=torch.float32) * 20 - 10)
          for _ in range(max(1, args.buffers))]
    taps = (torch.rand(2 * m, generator=gen, device=dev, dtype=torch.float32) * 2 - 1) / m
    y = torch.empty(2 * n, device=dev, dtype=torch.float32)     # convolution result
    scratch = torch.empty(2 * n, device=dev, dtype=torch.float32)  # FFT ping-pong partner
    stream = torch.cuda.current_stream().cuda_stream
    sp = C.c_void_p(stream)
    in_scratch = C.c_int(0)

    def step(i, ev=None):
        x = xs[i % len(xs)]
        # convolve_signal: ONE fused overlap-save launch; every workgroup transforms the zero-padded taps itself
        # before it starts on its blocks (no separate spectrum launch)
        if ev:
            lib.bdsp_hip_event_record(ev[0], sp)
        bd._lib.check(lib.bdsp_hip_dev_convolve(0, x.data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, sp))
        if ev:
            lib.bdsp_hip_event_record(ev[1], sp)
        # plain_fft of the filtered vector (3 Stockham passes at 2^24), y <-> scratch ping-pong
        bd._lib.check(lib.bdsp_hip_dev_fft(0, y.data_ptr(), scratch.data_ptr(), n, 1, 0, 1.0, -1,
                                           0.0, C.byref(in_scratch), sp))
        if ev:
            lib.bdsp_hip_event_record(ev[2], sp)

    # per-kernel durations come from HIP events inside the timed region; an event record costs about 2 us of stream
    # time, so only every `ev_stride`-th step carries the three events (at least eight steps do)
    ev_stride = max(1, min(8, args.steps // 8))
    events = {i: [lib.bdsp_hip_event_create() for _ in range(3)] for i in range(0, args.steps, ev_stride)}
    # Untimed clock pre-warm: the GPU idles at a few hundred MHz and needs tens of milliseconds of load to reach its
    # sustained clock (*measured*: the same step runs 233 us right after start-up and 216 us once the clock has
    # settled, tools/clock_probe.sh: 2.39 GHz, 1.37 kW under this kernel mix).  Then the W warm-up steps of the contract.
    t_pre = time.perf_counter()
    pre = 0
    while time.perf_counter() - t_pre < args.prewarm:
        for _ in range(25):
            step(pre)
            pre += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events.get(i))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # what an event pair costs by itself on this stream (two records with nothing between): the
    # per-kernel durations below are event deltas minus this, so they are comparable with rocprofv3's
    # kernel-only durations (profiles/*_bench_kernel_stats.csv)
    ms = C.c_float(0)
    empty = []
    for _ in range(20):
        ea, eb = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(ea, sp)
        lib.bdsp_hip_event_record(eb, sp)
        lib.bdsp_hip_synchronize(sp)
        lib.bdsp_hip_event_elapsed_ms(ea, eb, C.byref(ms))
        empty.append(ms.value)
        lib.bdsp_hip_event_destroy(ea)
        lib.bdsp_hip_event_destroy(eb)
    event_overhead = sorted(empty)[len(empty) // 2]

    conv_ms, fft_ms = [], []
    for e in events.values():
        lib.bdsp_hip_event_elapsed_ms(e[0], e[1], C.byref(ms))
        conv_ms.append(ms.value)
        lib.bdsp_hip_event_elapsed_ms(e[1], e[2], C.byref(ms))
        fft_ms.append(ms.value)
        for h in e:
            lib.bdsp_hip_event_destroy(h)
    conv_raw = sum(conv_ms) / len(conv_ms)
    conv_avg = max(conv_raw - event_overhead, 1e-6)
    fft_avg = max(sum(fft_ms) / len(fft_ms) - event_overhead, 1e-6)

    if rank == 0:
        samples = n * world * args.steps
        algo_bytes = 16.0 * n  # 8 B read + 8 B written per complex f32 sample
        dominant_conv = conv_avg >= fft_avg / 3.0  # compare one conv launch with one FFT pass
        achieved = algo_bytes / (conv_avg * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s for f32 complex FFT + overlap-save conv, 16M-pt",
            "value": samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "c3+fft16m: convolve_signal(%d-pt complex f32, %d complex taps, fused "
                            "overlap-save) -> plain_fft(%d-pt), one vector per GPU" % (n, m, n),
                "points": n, "taps": m, "vectors_per_gpu": 1, "input_buffers_rotated": len(xs),
                "untimed_clock_prewarm_s": args.prewarm, "untimed_prewarm_steps": pre,
                "steps_with_kernel_events": len(events),
                "parallelism": "independent vectors per GPU, no data-path collective",
            },
            "roofline": {
                "kernel": "k_overlap_save<float> (fused load->FFT4096->xH->IFFT4096->store)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic("k_overlap_save<float") if (n, m) == (POINTS, TAPS) else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_ms": conv_avg,
                "event_delta_ms": conv_raw, "event_pair_overhead_ms": event_overhead,
                "dominant": bool(dominant_conv),
            },
            "kernels": {
                "conv_ms": conv_avg, "conv_Msamples_s": n / (conv_avg * 1e-3) / 1e6,
                "fft_ms": fft_avg, "fft_Msamples_s": n / (fft_avg * 1e-3) / 1e6,
                "fft_passes": 3 if n > (1 << 20) else (2 if n > 4096 else 1),
                "fft_achieved_algorithmic_GBs": 16.0 * n / (fft_avg * 1e-3) / 1e9,
                "fft_achieved_pass_adjusted_GBs": 16.0 * n * (3 if n > (1 << 20) else 2) / (fft_avg * 1e-3) / 1e9,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, m, args.cpu_sample_points)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
