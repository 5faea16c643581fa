#!/usr/bin/env python3
"""bench.py -- headline benchmark of basic_dsp_amd (contract: see the build prompt / DESIGN.md 6).

Metric (BASELINE.json): Msamples/s for f32 complex FFT + overlap-save convolution at 16 777 216
points.  One "step" = one pass of the hot path over one synthetic vector per GPU:
    convolve_signal(x, 1024 complex taps)  ->  plain_fft        (config C3 + the 16M-point FFT)
`value` = complex samples that went through the whole step, summed over all ranks, per second,
with every buffer resident in HBM before the timed region.  Inputs rotate through several distinct
buffers so each step reads its input from HBM, not from the 256 MiB Infinity Cache.

Launch.  `python bench.py --gpus N` starts N ranks itself (one fresh child process per GPU, created
before this process has touched a GPU; the parent only waits and relays rank 0's JSON line) unless it
already runs under torchrun (WORLD_SIZE set), in which case --gpus must equal WORLD_SIZE.  Every rank
checks that all N ranks joined the RCCL group (`ranks_seen`); a short group is an error, not a
silently smaller benchmark.  Independent vectors per rank, no data-path collective (the path shards
by vector) -> weak scaling; only the timing barrier and the rank count use RCCL.

`--mode c5` is BASELINE config C5: 64 vectors of 1 048 576 points per GPU (512 on 8 GPUs), batched
convolve_signal -> plain_fft.  `value` is the compute-only rate with the shards resident; the same
line carries the end-to-end scatter + compute + gather time from rank 0 (chunked, pipelined:
basic_dsp_amd/batch.py) under `c5_end_to_end`.

Extra objects on the JSON line:
  roofline      the dominant kernel (the fused overlap-save launch): algorithmic bytes per launch
                (16 B per complex f32 sample) / mean launch duration measured with HIP
                events on the launch stream inside the timed region, against the 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (a port of the reference algorithm; the Rust reference itself cannot
                be built in this image) timed on a bounded sample of the same workload, rank 0, N=1:
                the reference's own schedule on one thread, the tail-free overlap-save on one thread,
                and the tail-free overlap-save on all host cores (numeric fields).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
POINTS = 1 << 24
TAPS = 1024
C5_POINTS = 1 << 20
C5_VECTORS_PER_GPU = 64


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=("headline", "c5"), default="headline")
    ap.add_argument("--prewarm", type=float, default=0.15, help="seconds of untimed load before the warm-up steps (clock ramp)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--taps", type=int, default=TAPS)
    ap.add_argument("--buffers", type=int, default=3, help="distinct input vectors rotated per step")
    ap.add_argument("--vectors-per-gpu", type=int, default=C5_VECTORS_PER_GPU, help="--mode c5")
    ap.add_argument("--chunk-vectors", type=int, default=8, help="--mode c5: vectors per pipelined scatter/gather chunk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-points", type=int, default=1 << 23)
    ap.add_argument("--dry-run-launch", action="store_true", help="print the per-rank child launches of --gpus N and exit")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """Parent side of `bench.py --gpus N` without torchrun: N children, one per LOCAL_RANK.  This process never
    imports torch or touches a GPU (a process that has initialised the GPU must not exec or fork workers)."""
    world = args.gpus
    port = _free_port()
    child_argv = [a for a in sys.argv[1:] if a != "--dry-run-launch"]
    launches = []
    for r in range(world):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
        launches.append({"cmd": [sys.executable, os.path.abspath(__file__)] + child_argv, "env": env})
    if args.dry_run_launch:
        for l in launches:
            print(json.dumps({"launch": l}))
        return 0
    procs = []
    for r, l in enumerate(launches):
        env = dict(os.environ)
        env.update(l["env"])
        procs.append(subprocess.Popen(l["cmd"], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        return 1
    return 0


# ------------------------------------------------------------------------------------------ CPU baseline
def usable_cores():
    """Host cores this process may actually run on: the affinity mask, cut down by a cgroup CPU quota if there is one
    (os.cpu_count() reports the machine, not the container)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(points, taps, sample_points):
    """The oracle on the host cores, on a bounded prefix of the workload (~10-20 s in all):
      reference_schedule_1core  overlap_discard exactly as the reference schedules it (scalar head, O(N*M/2) scalar
                                tail, blocks; convolution.rs:304-461) + the FFT, one thread = the reference's default
                                MultiCoreSettings (threading.rs:210-217)
      fair_1core                overlap-save with every output from a block (no scalar tail) + FFT, one thread
      fair_allcores             the same with blocks / butterflies spread over all host cores (OpenMP in the oracle;
                                the reference's `parallel()` setting would use half of them, threading.rs:220-231)
    `value` is the all-cores figure, `cores` the threads it used."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    n = min(points, sample_points)
    cores = usable_cores()
    x = orc.fill_uniform(2 * n, 201601171, -10, 10, np.float32)
    h = orc.fill_uniform(2 * taps, 201601172, -1, 1, np.float32) / np.float32(taps)
    l = orc.next_power_of_two(taps)
    t0 = time.perf_counter()
    code, y = orc.overlap_discard(x, h, l, fair=False)
    t1 = time.perf_counter()
    orc.fft(y)
    t2 = time.perf_counter()
    code2, y2 = orc.overlap_discard(x, h, l, fair=True)
    t3 = time.perf_counter()
    code3, y3 = orc.overlap_save_mt(x, h, l, cores)
    t4 = time.perf_counter()
    orc.fft_pow2_mt(y3, False, cores)
    t5 = time.perf_counter()
    assert code == 0 and code2 == 0 and code3 == 0 and np.array_equal(y2, y3)
    ref1 = n / (t2 - t0) / 1e6
    fair1 = n / ((t3 - t2) + (t2 - t1)) / 1e6
    fair_all = n / (t5 - t3) / 1e6
    return {
        "value": fair_all, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "%d-point prefix of the workload (convolve_signal with %d taps -> FFT), f32" % (n, taps),
        "machine_logical_cores": os.cpu_count(),
        "reference_schedule_1core_Msamples_s": ref1,
        "fair_1core_Msamples_s": fair1,
        "fair_allcores_Msamples_s": fair_all,
        "seconds": {"reference_overlap_discard_1core": t1 - t0, "fft_1core": t2 - t1, "fair_overlap_save_1core": t3 - t2,
                    "fair_overlap_save_allcores": t4 - t3, "fft_allcores": t5 - t4},
    }


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel as measured by rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
    runs, gfx950 read-size correction applied) -- collected with tools/pmc.sh on this same command and committed as
    profiles/r02_hbm_traffic.json; the live run cannot collect counters itself.  None if the profile is missing."""
    for name in ("r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                k = json.load(f)["kernels"]
            for kn, v in k.items():
                if kn.startswith(kernel_prefix):
                    return v["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
    return None


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    # TEST HOOK (tests/test_gpu_full_size.py): BDSP_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and runs the control
    # collectives over gloo on host tensors, so that the N-rank path (launcher, rendezvous, barriers, max over ranks, the
    # JSON line) can run on a ONE-GPU box.  RCCL refuses two ranks on one device; the line then says so in `config`.
    share_gpu = os.environ.get("BDSP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share_gpu else dev  # where the control collectives' tensors live
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        one = torch.ones(1, device=cdev, dtype=torch.int64)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        if ranks_seen != world:
            raise SystemExit("bench.py: %d of %d ranks joined the RCCL group" % (ranks_seen, world))
    import basic_dsp_amd as bd
    lib = bd.lib
    bd._lib.check(lib.bdsp_hip_set_device(local_rank), "set_device")
    bd.require_gpu()

    c5 = args.mode == "c5"
    n = args.points or (C5_POINTS if c5 else POINTS)
    m = args.taps
    nvec = args.vectors_per_gpu if c5 else 1
    gen = torch.Generator(device=dev)
    gen.manual_seed(201601171 + rank)
    # synthetic inputs, uniform(-10, 10) like tests/tools/mod.rs:124-139; taps uniform(-1,1)/M
    nbuf = max(1, args.buffers if not c5 else min(args.buffers, 2))
    xs = [(torch.rand(2 * n * nvec, generator=gen, device=dev, dtype=torch.float32) * 20 - 10) for _ in range(nbuf)]
    taps = (torch.rand(2 * m, generator=gen, device=dev, dtype=torch.float32) * 2 - 1) / m
    y = torch.empty(2 * n * nvec, device=dev, dtype=torch.float32)        # convolution result
    scratch = torch.empty(2 * n * nvec, device=dev, dtype=torch.float32)  # FFT ping-pong partner
    # torch's current stream; handle 0 (the default stream) is forwarded as HIP's null stream, not as "library stream"
    sp = bd._lib.torch_stream_arg()
    in_scratch = C.c_int(0)

    def step(i, ev=None):
        x = xs[i % len(xs)]
        # convolve_signal: ONE fused overlap-save launch over all vectors; every workgroup transforms the (delayed,
        # zero-padded) taps itself before it starts on its blocks
        if ev:
            lib.bdsp_hip_event_record(ev[0], sp)
        bd._lib.check(lib.bdsp_hip_dev_convolve(0, x.data_ptr(), y.data_ptr(), n, nvec, taps.data_ptr(), m, sp))
        if ev:
            lib.bdsp_hip_event_record(ev[1], sp)
        # plain_fft of the filtered vectors (3 Stockham passes at 2^24, 2 at 2^20), y <-> scratch ping-pong
        bd._lib.check(lib.bdsp_hip_dev_fft(0, y.data_ptr(), scratch.data_ptr(), n, nvec, 0, 1.0, -1, 0.0,
                                           C.byref(in_scratch), sp))
        if ev:
            lib.bdsp_hip_event_record(ev[2], sp)

    # per-kernel durations come from HIP events inside the timed region; an event record costs about 2 us of stream
    # time, so only every `ev_stride`-th step carries the three events (at least eight steps do)
    ev_stride = max(1, min(8, args.steps // 8))
    events = {i: [lib.bdsp_hip_event_create() for _ in range(3)] for i in range(0, args.steps, ev_stride)}
    # Untimed clock pre-warm: the GPU idles at a few hundred MHz and needs tens of milliseconds of load to reach its
    # sustained clock (*measured*: the same step runs 233 us right after start-up and 216 us once the clock has
    # settled).  Then the W warm-up steps of the contract.
    # ... but first the COLD figure a caller's first milliseconds see: three steps to load the code objects and fill the
    # workspace cache, a short idle, then twenty timed steps from the idle clock
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    time.sleep(0.05)
    t_cold = time.perf_counter()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t_cold) / 20 * 1e3
    t_pre = time.perf_counter()
    pre = 0
    while time.perf_counter() - t_pre < args.prewarm:
        for _ in range(25 if not c5 else 5):
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
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # what an event pair costs by itself on this stream (two records with nothing between): the per-kernel durations
    # below are event deltas minus this, so they are comparable with rocprofv3's kernel-only durations
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

    e2e = None
    if c5 and not (share_gpu and world > 1):  # (the chunked scatter/gather sends device tensors: RCCL only)
        e2e = c5_end_to_end(args, bd, torch, dist, dev, rank, world, n, m, nvec)

    if rank == 0:
        samples = n * nvec * world * args.steps
        algo_bytes = 16.0 * n * nvec  # 8 B read + 8 B written per complex f32 sample
        passes = 3 if n > (1 << 20) else (2 if n > 8192 else 1)
        achieved = algo_bytes / (conv_avg * 1e-3) / 1e9
        if c5:
            workload = ("c5: %d vectors of %d complex f32 points per GPU (%d in all), batched convolve_signal(%d taps, fused "
                        "overlap-save) -> plain_fft; value = compute only, shards resident" % (nvec, n, nvec * world, m))
        else:
            workload = ("c3+fft16m: convolve_signal(%d-pt complex f32, %d complex taps, fused overlap-save) -> "
                        "plain_fft(%d-pt), one vector per GPU" % (n, m, n))
        out = {
            "metric": "Msamples/s for f32 complex FFT + overlap-save conv, 16M-pt",
            "value": samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "points": n, "taps": m, "vectors_per_gpu": nvec, "input_buffers_rotated": len(xs),
                "untimed_clock_prewarm_s": args.prewarm, "untimed_prewarm_steps": pre,
                "cold_ms_per_step_first_20_steps_after_idle": cold_ms,
                "steps_with_kernel_events": len(events),
                "parallelism": "independent vectors per GPU, no data-path collective" + (
                    " -- TEST HOOK BDSP_BENCH_SHARE_GPU: all ranks on GPU 0, control collectives over gloo" if share_gpu else ""),
            },
            "roofline": {
                "kernel": "k_overlap_save_v2<R0=4> (fused load->FFT4096->xH->IFFT4096->store)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic("k_overlap_save_v2") if (n, m, nvec) == (POINTS, TAPS, 1) else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_ms": conv_avg,
                "event_delta_ms": conv_raw, "event_pair_overhead_ms": event_overhead,
                "dominant_per_launch": bool(conv_avg >= fft_avg / passes),
                "share_of_step": conv_avg / (conv_avg + fft_avg),
            },
            "kernels": {
                "conv_ms": conv_avg, "conv_Msamples_s": n * nvec / (conv_avg * 1e-3) / 1e6,
                "fft_ms": fft_avg, "fft_Msamples_s": n * nvec / (fft_avg * 1e-3) / 1e6,
                "fft_passes": passes,
                "fft_achieved_algorithmic_GBs": 16.0 * n * nvec / (fft_avg * 1e-3) / 1e9,
                "fft_achieved_pass_adjusted_GBs": 16.0 * n * nvec * passes / (fft_avg * 1e-3) / 1e9,
                "fft_frac_of_roofline_algorithmic": 16.0 * n * nvec / (fft_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "step_frac_of_roofline_algorithmic": 32.0 * n * nvec / ((conv_avg + fft_avg) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        if e2e is not None:
            out["c5_end_to_end"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n * nvec if c5 else n, m, args.cpu_sample_points)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def c5_end_to_end(args, bd, torch, dist, dev, rank, world, n, m, nvec):
    """BASELINE config C5 end to end: rank 0 holds all `nvec * world` vectors in its HBM, scatters them in chunks,
    every rank runs the batched kernels on its chunks while the next ones are in flight, the spectra travel back two
    rounds behind (basic_dsp_amd.batch.scatter_process_gather_chunked).  Reported next to the compute-only `value`
    (a shard is ~3.5 ms per direction on one xGMI link, ten times its compute time)."""
    from basic_dsp_amd.batch import process_shard_gpu, scatter_process_gather_chunked
    total = nvec * world
    batch = taps = None
    if rank == 0:
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        batch = torch.rand((total, 2 * n), generator=g, device=dev, dtype=torch.float32) * 20 - 10
        taps = (torch.rand(2 * m, generator=g, device=dev, dtype=torch.float32) * 2 - 1) / m
    times = []
    for it in range(4):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        out = scatter_process_gather_chunked(batch, taps, n, process_shard_gpu, chunk_vectors=args.chunk_vectors, device=dev)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        times.append(time.perf_counter() - t0)
        del out
    best = min(times[1:])
    return {"vectors": total, "chunk_vectors": args.chunk_vectors, "ms": best * 1e3,
            "Msamples_s": total * n / best / 1e6, "runs_ms": [t * 1e3 for t in times]}


def main():
    args = parse()
    under_launcher = "WORLD_SIZE" in os.environ
    if args.dry_run_launch or (not under_launcher and args.gpus is not None and args.gpus > 1):
        if args.gpus is None:
            args.gpus = 1
        sys.exit(launch_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
