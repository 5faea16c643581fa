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

Whenever the ranks form a process group (N > 1, or --init-dist) the DEFAULT mode also runs and VERIFIES the path's one
multi-GPU exchange after the timed region: a C5-shaped batch goes out from rank 0 in chunks over grouped point-to-point
sends, every rank convolves + transforms, the spectra come back, and rank 0 compares the first and last chunk of every
peer bit for bit with its own computation (`c5_end_to_end`: ms, Msamples_s, verified_rows, peers; a mismatch exits 3; a
leg that hangs is given up after --e2e-timeout and reported in the line).

Extra objects on the JSON line:
  value_windows the noise floor: window 0 is the contract's timed region (`value`), nine more windows of the same K steps
                follow back to back; min / median / max of the ten values and per-window kernel times, clock and power.
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
    ap.add_argument("--prewarm", type=float, default=0.3, help="seconds of untimed load before the warm-up steps (clock ramp; round 5: 0.15 -> 0.3 s, the ten 20-step windows of a run still climbed by 2 % after 0.15 s)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--taps", type=int, default=TAPS)
    ap.add_argument("--buffers", type=int, default=3, help="distinct input vectors rotated per step")
    ap.add_argument("--vectors-per-gpu", type=int, default=C5_VECTORS_PER_GPU, help="--mode c5")
    ap.add_argument("--chunk-vectors", type=int, default=8, help="--mode c5: vectors per pipelined scatter/gather chunk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--windows", type=int, default=9,
                    help="further timed windows of --steps steps after the contract's timed region (value_windows: the noise floor)")
    ap.add_argument("--e2e-vectors-per-gpu", type=int, default=8,
                    help="headline mode, ranks in a process group: vectors per GPU of the verified scatter/compute/gather leg")
    ap.add_argument("--e2e-chunk-vectors", type=int, default=2, help="vectors per pipelined chunk of that leg")
    ap.add_argument("--e2e-timeout", type=float, default=300.0,
                    help="seconds before a scatter/compute/gather leg that hangs is given up: the line is printed with an error in its place")
    ap.add_argument("--no-self-check", action="store_true",
                    help="skip the self-check of the timed step's output (for runs under rocprofv3: it launches torch kernels and "
                         "one more step)")
    ap.add_argument("--test-corrupt-self-check", action="store_true",
                    help="TEST HOOK, needs BDSP_BENCH_CORRUPT_SELF_CHECK=1 as well: one value of the checked convolution output is "
                         "changed before the self-check, which must then fail (exit code 5)")
    ap.add_argument("--no-first-call", action="store_true", help="skip the fresh-process first-call measurement (config.first_call_ms)")
    ap.add_argument("--cpu-sample-points", type=int, default=1 << 23)
    ap.add_argument("--dry-run-launch", action="store_true", help="print the per-rank child launches of --gpus N and exit")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="own launcher: seconds before the ranks are ended")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds a rank waits in the rendezvous / a collective")
    ap.add_argument("--init-dist", action="store_true",
                    help="join the RCCL process group even at world size 1 (every control collective then really runs "
                         "through librccl on the one GPU)")
    ap.add_argument("--test-share-gpu", action="store_true",
                    help="TEST HOOK, needs BDSP_BENCH_SHARE_GPU=1 as well: all ranks on GPU 0, control collectives over gloo; "
                         "the line is marked test_hook and carries no `value`")
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
        # (HSA_ENABLE_IPC_MODE_LEGACY=0 is set by every rank itself, run_rank(), so that this launcher and torchrun give
        # the ranks the same environment)
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
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
    failed, out0 = supervise(procs, args.launch_timeout)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: %s; exit codes %s\n" % (failed, [p.returncode for p in procs]))
        return 1
    return 0


def supervise(procs, timeout_s):
    """Waits for ALL rank processes at once.  procs[0].stdout is a pipe (rank 0 prints the JSON line) and is drained by
    a helper thread, so a full pipe cannot block rank 0 while this thread polls.  The first rank that exits non-zero --
    or the deadline -- ends the others at once (terminate, then kill) instead of leaving them in a rendezvous or a
    collective until the store / RCCL timeout, minutes later.  Returns (None or what went wrong, rank 0's output)."""
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % (bad[0], codes[bad[0]])
            # a peer that left because the multi-GPU leg failed on it (EXIT_E2E): rank 0 still owes the JSON line -- its own
            # collective fails or its watchdog fires within --e2e-timeout -- so it alone is waited for, up to the deadline
            if codes[0] is None and all(codes[r] == EXIT_E2E for r in bad) and time.monotonic() < deadline:
                time.sleep(0.05)
                continue
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = "no result after %.0f s (--launch-timeout)" % timeout_s
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5)
    return failed, b"".join(c for c in chunks if c)


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


def cpu_baseline(points, taps, sample_points, vectors=1, reps=3):
    """The oracle on the host cores, on a bounded sample of the workload (~15 s in all):
      reference_schedule_1core  overlap_discard exactly as the reference schedules it (scalar head, O(N*M/2) scalar
                                tail, blocks; convolution.rs:304-461) + the FFT, one thread = the reference's default
                                MultiCoreSettings (threading.rs:210-217)
      fair_1core                overlap-save with every output from a block (no scalar tail) + FFT, one thread
      fair_allcores             the same with blocks / butterflies spread over all host cores (OpenMP in the oracle;
                                the reference's `parallel()` setting would use half of them, threading.rs:220-231)
      numpy_sanity              scipy.signal.oaconvolve + numpy.fft.fft (pocketfft) on the same sample: NOT the reference --
                                it shows what a tuned CPU FFT does next to the port's radix-2 loop
    Headline: one `sample_points`-point prefix of the vector.  --mode c5 (vectors > 1): as many WHOLE vectors of
    `points` points as fit in `sample_points`, one after the other like the matrix crate's row loop
    (matrix/src/lib.rs:195-208).  Every leg runs `reps` times; the rates come from the MEDIAN time of a leg, the
    fastest repetition is reported next to it (`*_best`): the all-cores figure moved 57.9 -> 47.7 -> 42.9 Msamples/s over
    rounds 3-5 on unchanged code -- 16 granted cores of a shared 256-thread host -- and one sample could not bound that.
    `value` is the all-cores figure, `cores` the threads it used."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    if vectors > 1:
        n, k = points, max(1, min(vectors, sample_points // points))
    else:
        n, k = min(points, sample_points), 1
    cores = usable_cores()
    h = orc.fill_uniform(2 * taps, 201601172, -1, 1, np.float32) / np.float32(taps)
    l = orc.next_power_of_two(taps)
    legs = ("reference_overlap_discard_1core", "fft_1core", "fair_overlap_save_1core", "fair_overlap_save_allcores", "fft_allcores",
            "numpy_oaconvolve", "numpy_fft")
    runs = {key: [] for key in legs}
    xs = [orc.fill_uniform(2 * n, 201601171 + v, -10, 10, np.float32) for v in range(k)]
    try:
        import scipy.signal as ss
    except ImportError:
        ss = None
    hc = h.view(np.complex64)
    for rep in range(max(1, reps)):
        sec = {key: 0.0 for key in legs}
        for x in xs:
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
            t6 = t7 = t5
            if ss is not None:
                yn = ss.oaconvolve(x.view(np.complex64), hc, "same")
                t6 = time.perf_counter()
                np.fft.fft(yn)
                t7 = time.perf_counter()
            for key, dt in zip(legs, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)):
                sec[key] += dt
        for key in legs:
            runs[key].append(sec[key])
    med = {key: sorted(v)[len(v) // 2] for key, v in runs.items()}
    best = {key: min(v) for key, v in runs.items()}
    tot = n * k

    def rate(t, a, b):
        return tot / (t[a] + t[b]) / 1e6 if t[a] + t[b] > 0 else None
    if vectors > 1:
        sample = "%d whole vectors of %d points of the batch (convolve_signal with %d taps -> FFT each), f32" % (k, n, taps)
    else:
        sample = "%d-point prefix of the workload (convolve_signal with %d taps -> FFT), f32" % (n, taps)
    fair_all = rate(med, "fair_overlap_save_allcores", "fft_allcores")
    return {
        "value": fair_all, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": sample,
        "machine_logical_cores": os.cpu_count(),
        "reps": max(1, reps), "statistic": "median of the repetitions (fastest repetition: *_best)",
        "reference_schedule_1core_Msamples_s": rate(med, "reference_overlap_discard_1core", "fft_1core"),
        "fair_1core_Msamples_s": rate(med, "fair_overlap_save_1core", "fft_1core"),
        "fair_allcores_Msamples_s": fair_all,
        "reference_schedule_1core_Msamples_s_best": rate(best, "reference_overlap_discard_1core", "fft_1core"),
        "fair_1core_Msamples_s_best": rate(best, "fair_overlap_save_1core", "fft_1core"),
        "fair_allcores_Msamples_s_best": rate(best, "fair_overlap_save_allcores", "fft_allcores"),
        # pocketfft / scipy on one thread: a sanity row, not the reference and not the port
        "numpy_sanity_Msamples_s": rate(med, "numpy_oaconvolve", "numpy_fft") if ss is not None else None,
        "numpy_sanity": "scipy.signal.oaconvolve + numpy.fft.fft (pocketfft), library default threading -- NOT the reference",
        "seconds": med, "seconds_runs": runs,
    }


PROFILE_TAG = "r06"  # profiles/<tag>_* are the files tools/profile_round.sh writes (TAG=r06)


def kernel_source_sha16():
    """Hash of the kernel sources the profiles were collected on (tools/profile_round.sh stores it in
    profiles/<tag>_profile_meta.json): figures from a profile of OTHER sources are not quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "basic_dsp_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_figures():
    """What the committed rocprofv3 runs of THIS command say (tools/profile_round.sh): HBM bytes per launch from the PMC
    passes (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 read-size correction applied) and the kernel-trace
    average durations.  The live run cannot collect counters itself.  Everything is None when the profile is missing
    or was collected on other kernel sources than the ones in this tree."""
    out = {"conv_traffic": None, "fft_traffic": None, "conv_avg_ns": None, "conv_min_ns": None, "fft_avg_ns": None, "stale": None}
    pdir = os.path.join(ROOT, "profiles")
    try:
        with open(os.path.join(pdir, PROFILE_TAG + "_profile_meta.json")) as f:
            meta = json.load(f)
    except (OSError, ValueError):
        return out
    out["stale"] = meta.get("source_sha16") != kernel_source_sha16()
    if out["stale"]:
        return out
    try:
        with open(os.path.join(pdir, PROFILE_TAG + "_hbm_traffic.json")) as f:
            k = json.load(f)["kernels"]
        conv = [v["hbm_bytes_per_launch"] for kn, v in k.items() if kn.startswith("k_overlap_save_v2")]
        fft = [v["hbm_bytes_per_launch"] for kn, v in k.items() if kn.startswith("k_fft_pass")]
        if conv:
            out["conv_traffic"] = conv[0]
        if len(fft) == 2:  # the first-pass instantiation runs once per transform, the later-pass one twice (3 passes)
            first = [v["hbm_bytes_per_launch"] for kn, v in k.items() if kn.startswith("k_fft_pass") and ", -1, true, false, true" in kn]
            later = [v["hbm_bytes_per_launch"] for kn, v in k.items() if kn.startswith("k_fft_pass") and ", -1, false, false, true" in kn]
            if first and later:
                out["fft_traffic"] = first[0] + 2 * later[0]
    except (OSError, KeyError, ValueError):
        pass
    try:
        import csv
        with open(os.path.join(pdir, PROFILE_TAG + "_bench_kernel_stats.csv")) as f:
            fft_ns = 0.0
            for row in csv.DictReader(f):
                name = row["Name"]
                if "bdsp::k_overlap_save_v2" in name:
                    out["conv_avg_ns"] = float(row["AverageNs"])
                    out["conv_min_ns"] = float(row["MinNs"])
                elif "bdsp::k_fft_pass<float, 256, 16, -1, true" in name:
                    fft_ns += float(row["AverageNs"])
                elif "bdsp::k_fft_pass<float, 256, 16, -1, false" in name:
                    fft_ns += 2 * float(row["AverageNs"])
            out["fft_avg_ns"] = fft_ns or None
    except (OSError, KeyError, ValueError):
        pass
    return out



# ------------------------------------------------------------------------------------------ clock / power samples
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy on MI355X, /opt/skills/guides/MI355X_MICROARCH.md (secondary fraction)


class GpuSampler:
    """Host-side samples of the GPU's core clock and socket power while the benchmark runs: a thread polls
    librocm_smi64 (sysfs reads, nothing is launched on the GPU) every few milliseconds; `window(a, b)` summarises the
    samples taken between two perf_counter times.  Absent library / device / permission: every figure is None --
    the bench line then says nothing about clocks instead of repeating an old measurement."""

    def __init__(self, pci_bus=None, pci_device=None, period_s=0.0005):
        self.samples = []  # (t, sclk_mhz or None, power_w or None)
        self.cap_w = None
        self.sclk_max_mhz = None
        self.source = None
        self._stop = False
        self._thread = None
        self._period = period_s
        try:
            lib = C.CDLL("librocm_smi64.so")

            class Freqs(C.Structure):
                _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32),
                            ("frequency", C.c_uint64 * 33)]
            self._Freqs = Freqs
            if lib.rsmi_init(C.c_uint64(0)) != 0:
                return
            n = C.c_uint32(0)
            if lib.rsmi_num_monitor_devices(C.byref(n)) != 0 or n.value == 0:
                return
            dv = 0
            if n.value > 1 and pci_bus is not None:
                for i in range(n.value):
                    bdf = C.c_uint64(0)
                    if lib.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(bdf)) == 0 and ((bdf.value >> 8) & 0xff) == pci_bus and \
                            (pci_device is None or ((bdf.value >> 3) & 0x1f) == pci_device):
                        dv = i
                        break
            self._lib, self._dv = lib, C.c_uint32(dv)
            cap = C.c_uint64(0)
            if lib.rsmi_dev_power_cap_get(self._dv, C.c_uint32(0), C.byref(cap)) == 0 and cap.value:
                self.cap_w = cap.value / 1e6
            f = Freqs()
            if lib.rsmi_dev_gpu_clk_freq_get(self._dv, C.c_int(0), C.byref(f)) == 0 and 0 < f.num_supported <= 33:
                self.sclk_max_mhz = max(f.frequency[i] for i in range(f.num_supported)) / 1e6
            self.source = "librocm_smi64 (rsmi_dev_gpu_clk_freq_get SYS, rsmi_dev_power_get), device %d of %d" % (dv, n.value)
        except (OSError, AttributeError):
            self.source = None

    def _read(self):
        sclk = power = None
        f = self._Freqs()
        if self._lib.rsmi_dev_gpu_clk_freq_get(self._dv, C.c_int(0), C.byref(f)) == 0 and f.current < 33:
            sclk = f.frequency[f.current] / 1e6
        pw, ty = C.c_uint64(0), C.c_int(0)
        try:
            if self._lib.rsmi_dev_power_get(self._dv, C.byref(pw), C.byref(ty)) == 0:
                power = pw.value / 1e6
        except AttributeError:
            if self._lib.rsmi_dev_current_socket_power_get(self._dv, C.byref(pw)) == 0:
                power = pw.value / 1e6
        return sclk, power

    def start(self):
        if self.source is None:
            return self
        import threading

        def loop():
            while not self._stop:
                t = time.perf_counter()
                try:
                    sclk, power = self._read()
                except Exception:  # noqa: BLE001  (a sampler must never take the benchmark down)
                    break
                self.samples.append((t, sclk, power))
                time.sleep(self._period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop = True
        if self._thread:
            self._thread.join(timeout=1)

    def window(self, a, b):
        """Median clock / power of the samples with a <= t <= b (None when there are none)."""
        def med(v):
            v = sorted(x for x in v if x is not None)
            return v[len(v) // 2] if v else None
        w = [x for x in self.samples if a <= x[0] <= b]
        return {"samples": len(w), "sclk_mhz": med([x[1] for x in w]), "socket_power_w": med([x[2] for x in w])}


FIRST_CALL_CHILD = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import basic_dsp_amd as bd
from basic_dsp_amd import DspVec
n, m = int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1)
x = (rng.random(2 * n, dtype=np.float32) * 20 - 10)
h = ((rng.random(2 * m, dtype=np.float32) * 2 - 1) / m).astype(np.float32)
t = time.perf_counter(); bd.require_gpu(); v = DspVec(x, is_complex=True); hv = DspVec(h, is_complex=True); w = DspVec(x, is_complex=True)
bd.lib.bdsp_hip_synchronize(None); up = time.perf_counter() - t
out = {"device_init_and_upload_ms": up * 1e3}
for name, vec, call in (("plain_fft", v, lambda q: q.plain_fft()), ("convolve_signal", w, lambda q: q.convolve_signal(hv))):
    for rep in ("first", "second", "third"):
        t = time.perf_counter(); assert call(vec) == 0; bd.lib.bdsp_hip_synchronize(None)
        out[name + "_" + rep + "_ms"] = (time.perf_counter() - t) * 1e3
        if name == "plain_fft": vec.plain_ifft()  # back to the time domain (untimed); its own first call loads nothing new but tables
        bd.lib.bdsp_hip_synchronize(None)
import json; print("FIRSTCALL " + json.dumps(out))
"""


def first_call_cost(points, taps):
    """What a caller's very FIRST plain_fft / convolve_signal cost in a fresh process (code-object load of the 570-kernel
    library, twiddle tables, workspace allocation) next to the second and third call of the same process.  Runs in a
    child process (numpy + ctypes, no torch) after the timed region; None when it fails."""
    try:
        p = subprocess.run([sys.executable, "-c", FIRST_CALL_CHILD, ROOT, str(points), str(taps)], capture_output=True, text=True, timeout=300)
        for line in p.stdout.splitlines():
            if line.startswith("FIRSTCALL "):
                return json.loads(line[len("FIRSTCALL "):])
        return {"error": (p.stderr or p.stdout)[-300:]}
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"error": str(e)[-300:]}

# ------------------------------------------------------------------------------------------ self-check of the timed output
EXIT_SELF_CHECK, EXIT_E2E = 5, 4  # after the JSON line is out: the step's output is wrong / the multi-GPU leg hung or failed


def dft_bins(yc, ks):
    """X[k] = sum_n y[n] exp(-2 pi i n k / N) for a few k, from the definition, in f64 on the host: n = a B + b splits the
    phase into two small tables of EXACT phases (integer k n mod N), the sum into one matrix product -- 16M points x 6 bins
    cost 0.1 s instead of 16M complex exponentials per bin (the arithmetic of tests/test_gpu_full_size.py:85-91)."""
    import numpy as np
    n = yc.shape[0]
    b = 1
    while b * b < n:
        b <<= 1
    while n % b:
        b >>= 1
    a = n // b
    ks = np.asarray(ks, dtype=np.int64)
    ib, ia = np.arange(b, dtype=np.int64), np.arange(a, dtype=np.int64)
    tb = np.exp(-2j * np.pi * ((ib[:, None] * ks[None, :]) % n) / n)                 # (b, K)
    ta = np.exp(-2j * np.pi * (((ia[:, None] * b) % n * ks[None, :]) % n) / n)       # (a, K)
    return np.sum((yc.reshape(a, b) @ tb) * ta, axis=0)


def self_check(torch, np, orc, x, y_t, spec_t, n, m, taps_t, corrupt=False):
    """Checks ONE vector of the last timed step: x (input), y_t (its convolve_signal output, recomputed bit-identically
    after the timed region because the 3-pass transform overwrites it), spec_t (the TIMED step's spectrum).  The checker is
    the CPU oracle (tests/oracle_lib.py, f64) and numpy -- never the library under test.
      conv_rel_l2_max  three 4 096-output windows of y (vector start and end -- both read across the wrap-around -- and the
                       middle) against orc.convolve_direct in f64 (time_freq/mod.rs:455-473)
      fft_bin_err_max  six bins of the spectrum against the DFT definition of y in f64, relative to sqrt(N) rms|y|
      parseval_rel     | sum|X|^2 / N - sum|y|^2 | / sum|y|^2 in f64
    Tolerances: north_star's 1e-6 (2e-6 for a single bin, as in tests/test_gpu_full_size.py)."""
    t0 = time.perf_counter()
    W, pad = 4096, ((m + 1) // 2 + 63) // 64 * 64 + 64
    h64 = taps_t.cpu().numpy().astype(np.float64)
    if corrupt:
        y_t[2 * (n // 2) + 200] += 1.0
    conv_err = 0.0
    firsts = sorted({0, max(0, n // 2 - W // 2), max(0, n - W)}) if n >= W + 2 * pad else [None]
    for first in firsts:
        if first is None:  # a vector shorter than a window + margins: the whole of it
            ref = orc.convolve_direct(x.cpu().numpy().astype(np.float64), h64, True)
            got = y_t.cpu().numpy()
        else:
            idx = (torch.arange(first - pad, first + W + pad, device=x.device) % n)
            xs = x.view(-1, 2)[idx].reshape(-1).cpu().numpy().astype(np.float64)
            # the oracle convolves circularly over what it is given; outputs `pad` away from the ends of this excerpt never
            # see its seam, so they are the outputs first .. first + W of the whole vector
            ref = orc.convolve_direct(xs, h64, True, pad, W)
            got = y_t[2 * first:2 * (first + W)].cpu().numpy()
        conv_err = max(conv_err, float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)))
    yc = y_t.cpu().numpy().view(np.complex64).astype(np.complex128)
    X = spec_t.cpu().numpy().view(np.complex64)
    ks = sorted({0, 1, min(4097, n - 1), n // 2, n - 1, 12_345_678 % n})
    ref = dft_bins(yc, ks)
    e_t = float(np.vdot(yc, yc).real)
    scale = np.sqrt(e_t)  # = sqrt(N) * rms|y|: the typical magnitude of a bin
    bin_err = float(np.max(np.abs(X[ks].astype(np.complex128) - ref)) / max(scale, 1e-300))
    Xr = X.view(np.float32).astype(np.float64)
    e_f = float(np.dot(Xr, Xr)) / n
    parseval = abs(e_f - e_t) / max(e_t, 1e-300)
    ok = bool(conv_err < 1e-6 and bin_err < 2e-6 and parseval < 1e-6)
    return {"ok": ok, "conv_rel_l2_max": conv_err, "fft_bin_err_max": bin_err, "parseval_rel": parseval,
            "conv_windows": ["whole vector"] if firsts == [None] else [[f, f + W] for f in firsts], "fft_bins": [int(k) for k in ks],
            "tolerances": {"conv_rel_l2": 1e-6, "fft_bin": 2e-6, "parseval": 1e-6},
            "checker": "CPU oracle orc.convolve_direct (f64) + DFT definition / Parseval in numpy f64; product library not involved",
            "seconds": time.perf_counter() - t0}


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    # the host driver only supports dmabuf IPC; set here (before torch loads HSA) so that the own launcher and torchrun
    # give every rank the same environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    # TEST HOOK (tests/test_gpu_full_size.py): --test-share-gpu TOGETHER WITH BDSP_BENCH_SHARE_GPU=1 puts every rank on
    # GPU 0 and runs the control collectives over gloo on host tensors, so that the N-rank path (launcher, rendezvous,
    # barriers, max over ranks, the JSON line) can run on a ONE-GPU box.  RCCL refuses two ranks on one device.  The
    # line is then marked `test_hook` and its `value` is null: a leaked environment variable alone changes nothing.
    share_gpu = args.test_share_gpu and os.environ.get("BDSP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share_gpu else dev  # where the control collectives' tensors live
    ranks_seen = 1
    use_dist = world > 1 or args.init_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if share_gpu:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        one = torch.ones(1, device=cdev, dtype=torch.int64)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        if ranks_seen != world:
            raise SystemExit("bench.py: %d of %d ranks joined the RCCL group" % (ranks_seen, world))
    import basic_dsp_amd as bd
    lib = bd.lib
    bd._lib.check(lib.bdsp_hip_set_device(local_rank), "set_device")
    bd.require_gpu()

    props = torch.cuda.get_device_properties(local_rank)
    smi = GpuSampler(getattr(props, "pci_bus_id", None), getattr(props, "pci_device_id", None)).start() if rank == 0 else None

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
    # time, so only every `ev_stride`-th step carries the three events (at least five steps do)
    ev_stride = max(1, min(8, args.steps // 5))
    ms = C.c_float(0)

    def timed_window():
        """EXACTLY --steps steps between barrier + synchronize on both sides (the contract's timed region; the further
        windows of `value_windows` repeat it).  Returns (seconds on this rank, t0, t_end, event deltas conv / fft in ms)."""
        events = {i: [lib.bdsp_hip_event_create() for _ in range(3)] for i in range(0, args.steps, ev_stride)}
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, events.get(i))
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        cm, fm = [], []
        for e in events.values():
            lib.bdsp_hip_event_elapsed_ms(e[0], e[1], C.byref(ms))
            cm.append(ms.value)
            lib.bdsp_hip_event_elapsed_ms(e[1], e[2], C.byref(ms))
            fm.append(ms.value)
            for h in e:
                lib.bdsp_hip_event_destroy(h)
        return el, t0, t0 + el, cm, fm

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
    t_pre_end = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # ---- the contract's timed region: `value`, `ms_per_step`, `roofline` come from THIS window only
    elapsed, t0, t_end, conv_ms, fft_ms = timed_window()
    # ---- the noise floor (round 5): further windows of the same K steps, back to back, same barriers.  r03 -> r04 moved
    # the headline by -2.9 % on identical kernel sources and the record could not say whether that was noise.
    more = [timed_window() for _ in range(max(0, args.windows))]
    win_elapsed = [elapsed] + [w[0] for w in more]
    own_elapsed = [elapsed]  # every rank's OWN window-0 time (value_by_rank: a straggling GPU must be visible in a weak-scaling sum)
    if use_dist:
        t = torch.tensor(win_elapsed, device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        own = torch.zeros(world, device=cdev, dtype=torch.float64)
        own[rank] = elapsed
        dist.all_reduce(own, op=dist.ReduceOp.SUM)
        own_elapsed = [float(v) for v in own.tolist()]
        win_elapsed = [float(v) for v in t.tolist()]
        elapsed = win_elapsed[0]
    steps_with_events = len(conv_ms)

    # what an event pair costs by itself on this stream (two records with nothing between): the per-kernel durations
    # below are event deltas minus this, so they are comparable with rocprofv3's kernel-only durations
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
    # (the smallest of the twenty: a pair recorded on a stream that has just gone idle can cost several times the 4-5 us a
    # pair costs between kernels -- *measured* 18 us medians after a 200-step run -- and would be over-subtracted)
    event_overhead = min(empty)

    conv_raw = sum(conv_ms) / len(conv_ms)
    conv_avg = max(conv_raw - event_overhead, 1e-6)
    fft_avg = max(sum(fft_ms) / len(fft_ms) - event_overhead, 1e-6)

    # ---- the self-check (round 6): the line certifies the output of the step it timed.  The last step of the last window
    # read xs[(K-1) % buffers]; its spectrum is still in y / scratch.  The convolution result it transformed is gone (the
    # passes ping-pong through y), so that step runs ONCE more, untimed, on the same input: its spectrum must equal the
    # timed one bit for bit (same kernels, same data), and its y is what the checker sees.
    check = None
    if rank == 0 and not args.no_self_check:
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as orc
        t_sc = time.perf_counter()
        last = args.steps - 1
        torch.cuda.synchronize()
        spec_timed = (scratch if in_scratch.value else y).clone()
        x_last = xs[last % len(xs)]
        bd._lib.check(lib.bdsp_hip_dev_convolve(0, x_last.data_ptr(), y.data_ptr(), n, nvec, taps.data_ptr(), m, sp))
        torch.cuda.synchronize()
        y_keep = y.clone()
        bd._lib.check(lib.bdsp_hip_dev_fft(0, y.data_ptr(), scratch.data_ptr(), n, nvec, 0, 1.0, -1, 0.0, C.byref(in_scratch), sp))
        torch.cuda.synchronize()
        same = bool(torch.equal(spec_timed, scratch if in_scratch.value else y))
        vsel = nvec - 1  # the batch's last vector
        sl = slice(2 * n * vsel, 2 * n * (vsel + 1))
        corrupt = args.test_corrupt_self_check and os.environ.get("BDSP_BENCH_CORRUPT_SELF_CHECK") == "1"
        check = self_check(torch, np, orc, x_last[sl], y_keep[sl], spec_timed[sl], n, m, taps, corrupt)
        check["rerun_spectrum_bit_identical_to_timed"] = same
        check["ok"] = bool(check["ok"] and same)
        check["vector_checked"] = vsel
        check["seconds"] = time.perf_counter() - t_sc
        del spec_timed, y_keep

    out = None
    if rank == 0:
        samples = n * nvec * world * args.steps
        win_values = [samples / e / 1e6 for e in win_elapsed]
        sv = sorted(win_values)
        win_spans = [(t0, t_end)] + [(w[1], w[2]) for w in more]
        win_conv = [conv_ms] + [w[3] for w in more]
        win_fft = [fft_ms] + [w[4] for w in more]
        algo_bytes = 16.0 * n * nvec  # 8 B read + 8 B written per complex f32 sample
        passes = int(lib.bdsp_hip_fft_passes(0, n)) or 1  # what bdsp_hip_dev_fft launches for this length
        achieved = algo_bytes / (conv_avg * 1e-3) / 1e9
        headline = (n, m, nvec) == (POINTS, TAPS, 1)
        prof = profile_figures() if headline else {}
        fft_algo_gbs = 16.0 * n * nvec / (fft_avg * 1e-3) / 1e9
        step_algo_gbs = 32.0 * n * nvec / ((conv_avg + fft_avg) * 1e-3) / 1e9
        if smi is not None:
            smi.stop()
            w_pre, w_timed = smi.window(t_pre, t_pre_end), smi.window(t0, t_end)
            clk = {"source": smi.source, "prewarm": w_pre, "timed": w_timed, "power_cap_w": smi.cap_w, "sclk_max_mhz": smi.sclk_max_mhz,
                   "limited_by": None}
            pw = w_timed["socket_power_w"] if w_timed["socket_power_w"] is not None else w_pre["socket_power_w"]
            ck = w_timed["sclk_mhz"] if w_timed["sclk_mhz"] is not None else w_pre["sclk_mhz"]
            if pw is not None and ck is not None and smi.cap_w and smi.sclk_max_mhz:
                if pw >= 0.95 * smi.cap_w and ck <= 0.92 * smi.sclk_max_mhz:
                    clk["limited_by"] = "socket power cap: %.0f of %.0f W, sclk %.0f of %.0f MHz (sampled in this run)" % (pw, smi.cap_w, ck, smi.sclk_max_mhz)
                elif ck <= 0.92 * smi.sclk_max_mhz:
                    clk["limited_by"] = "sclk %.0f of %.0f MHz at %.0f of %.0f W (sampled in this run)" % (ck, smi.sclk_max_mhz, pw, smi.cap_w)
        else:
            clk = {"source": None, "prewarm": {"samples": 0, "sclk_mhz": None, "socket_power_w": None},
                   "timed": {"samples": 0, "sclk_mhz": None, "socket_power_w": None}, "power_cap_w": None, "sclk_max_mhz": None, "limited_by": None}
        if fft_avg >= conv_avg:
            dominant_by_time = {"kernel": "k_fft_pass<float,256,16> x %d passes (plain_fft)" % passes if not c5 else "k_fft_pass x %d passes (plain_fft)" % passes,
                                "share_of_step": fft_avg / (conv_avg + fft_avg), "ms": fft_avg,
                                "frac": fft_algo_gbs / HBM_PEAK_GBS, "frac_vs_achievable": fft_algo_gbs / HBM_ACHIEVABLE_GBS,
                                "frac_pass_adjusted": fft_algo_gbs * passes / HBM_PEAK_GBS,
                                "traffic_ratio": (prof["fft_traffic"] / (16.0 * n * nvec)) if prof.get("fft_traffic") else None}
        else:
            dominant_by_time = {"kernel": "k_overlap_save_v2 (convolve_signal)", "share_of_step": conv_avg / (conv_avg + fft_avg), "ms": conv_avg,
                                "frac": achieved / HBM_PEAK_GBS, "frac_vs_achievable": achieved / HBM_ACHIEVABLE_GBS,
                                "frac_pass_adjusted": achieved / HBM_PEAK_GBS, "traffic_ratio": None}
        first_call = None
        if world == 1 and not c5 and not args.no_first_call:
            first_call = first_call_cost(n, m)
        if c5:
            workload = ("c5: %d vectors of %d complex f32 points per GPU (%d in all), batched convolve_signal(%d taps, fused "
                        "overlap-save) -> plain_fft; value = compute only, shards resident" % (nvec, n, nvec * world, m))
        else:
            workload = ("c3+fft16m: convolve_signal(%d-pt complex f32, %d complex taps, fused overlap-save) -> "
                        "plain_fft(%d-pt), one vector per GPU" % (n, m, n))
        out = {
            "metric": ("Msamples/s for f32 complex FFT + overlap-save conv, 16M-pt" if not c5 else
                       "Msamples/s for f32 complex FFT + overlap-save conv, batch of %d x %d-pt vectors per GPU (BASELINE config C5)" % (nvec, n)),
            "value": None if share_gpu else samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            # each rank's own rate over window 0 (its own clock between the same two barriers); `value` divides ALL ranks'
            # samples by the slowest rank's time
            "value_by_rank": None if share_gpu else [n * nvec * args.steps / e / 1e6 for e in own_elapsed],
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
                # a fresh process's first plain_fft / convolve_signal against its second and third
                "first_call_ms": first_call,
                "steps_with_kernel_events": steps_with_events,
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
                "traffic": prof.get("conv_traffic"),
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_ms": conv_avg,
                "event_delta_ms": conv_raw, "event_pair_overhead_ms": event_overhead,
                # clock and socket power of THIS run (host thread polling librocm_smi64 during the untimed pre-warm -- the
                # same load as the timed steps, tens of samples -- and during the timed region itself); `limited_by` is a
                # sentence DERIVED from them or null, never a remembered measurement
                "sclk_mhz": clk["timed"]["sclk_mhz"] if clk["timed"]["sclk_mhz"] is not None else clk["prewarm"]["sclk_mhz"],
                "socket_power_w": clk["timed"]["socket_power_w"] if clk["timed"]["socket_power_w"] is not None else clk["prewarm"]["socket_power_w"],
                "power_cap_w": clk["power_cap_w"], "sclk_max_mhz": clk["sclk_max_mhz"],
                "clock_power_samples": clk,
                "limited_by": clk["limited_by"],
                # the same achieved rate against the 6.29 TB/s a float4 copy reaches on this chip
                "frac_vs_achievable": achieved / HBM_ACHIEVABLE_GBS,
                "dominant_per_launch": bool(conv_avg >= fft_avg / passes),
                "share_of_step": conv_avg / (conv_avg + fft_avg),
                # the kernel with the largest share of the STEP and ITS fraction, so that `frac` above (the fused block
                # kernel, the dominant kernel per launch) cannot be read as the step's
                "dominant_by_time": dominant_by_time,
                # the whole step and the transform, in the same terms (algorithmic bytes: 32 B per sample for the step,
                # 16 B per sample for the transform whatever its number of passes) -- the block kernel is a third of the step
                "step_frac": step_algo_gbs / HBM_PEAK_GBS,
                "step_frac_vs_achievable": step_algo_gbs / HBM_ACHIEVABLE_GBS,
                "fft_frac_algorithmic": fft_algo_gbs / HBM_PEAK_GBS,
                "fft_passes": passes,
                # HBM bytes the transform's passes moved per algorithmic byte (PMC, profiles/) -- 3.0 = three full trips
                "fft_traffic_ratio": (prof["fft_traffic"] / (16.0 * n * nvec)) if prof.get("fft_traffic") else None,
                # the same fraction from the committed rocprofv3 kernel trace of this command (average over ALL launches
                # incl. the clock ramp and the cold steps, and the fastest launch)
                "frac_rocprof": (algo_bytes / (prof["conv_avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBS) if prof.get("conv_avg_ns") else None,
                "frac_rocprof_min_launch": (algo_bytes / (prof["conv_min_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBS) if prof.get("conv_min_ns") else None,
                "profile": ("profiles/%s_* (%s)" % (PROFILE_TAG, "collected on these kernel sources" if prof.get("stale") is False else
                            ("STALE: collected on other kernel sources, figures withheld" if prof.get("stale") else "missing"))) if headline else None,
            },
            "kernels": {
                "conv_ms": conv_avg, "conv_Msamples_s": n * nvec / (conv_avg * 1e-3) / 1e6,
                "fft_ms": fft_avg, "fft_Msamples_s": n * nvec / (fft_avg * 1e-3) / 1e6,
                "fft_passes": passes,
                "fft_achieved_algorithmic_GBs": fft_algo_gbs,
                "fft_achieved_pass_adjusted_GBs": fft_algo_gbs * passes,
                "fft_frac_of_roofline_algorithmic": fft_algo_gbs / HBM_PEAK_GBS,
                "step_frac_of_roofline_algorithmic": step_algo_gbs / HBM_PEAK_GBS,
            },
        }
        # the noise floor: window 0 IS the contract's timed region (`value`), windows 1.. repeat it back to back
        wclk = [smi.window(a, b) if smi is not None else {"samples": 0, "sclk_mhz": None, "socket_power_w": None} for a, b in win_spans]
        out["value_windows"] = {
            "n": len(win_values), "steps_per_window": args.steps, "first_is_value": True,
            "min": None if share_gpu else sv[0], "median": None if share_gpu else sv[len(sv) // 2], "max": None if share_gpu else sv[-1],
            "spread_pct": None if share_gpu else (sv[-1] - sv[0]) / sv[len(sv) // 2] * 100.0,
            "values": None if share_gpu else win_values,
            "ms_per_step": [e / args.steps * 1e3 for e in win_elapsed],
            "conv_ms": [max(sum(c) / len(c) - event_overhead, 1e-6) for c in win_conv],
            "fft_ms": [max(sum(f) / len(f) - event_overhead, 1e-6) for f in win_fft],
            "sclk_mhz": [w["sclk_mhz"] for w in wclk], "socket_power_w": [w["socket_power_w"] for w in wclk],
            "clock_power_samples": [w["samples"] for w in wclk],
            "sampler_period_ms": 0.5,
        }
        out["self_check"] = check
        if share_gpu:
            out["test_hook"] = True
            out["test_hook_value"] = samples / elapsed / 1e6

    # ---- the path's multi-GPU exchange, AFTER every figure of the line above is final and under a watchdog: a leg that
    # hangs in a collective must not cost the record the measurements that were complete before it started
    def e2e_leg():
        if c5 and not (share_gpu and world > 1):  # (the chunked scatter/gather sends device tensors: RCCL only)
            return c5_end_to_end(args, bd, torch, dist, dev, rank, world, n, m, nvec, use_dist)
        if not c5 and use_dist:
            # the headline mode is what the driver's scaling runs launch (`bench.py --gpus N`): its ranks work on independent
            # vectors, so without this leg an N-GPU record would never execute the path's one multi-GPU exchange
            return verified_scatter_gather(args, torch, dist, dev, rank, world, m, share_gpu)
        return None

    e2e, group_ok = guarded_leg(e2e_leg, args.e2e_timeout, rank, out)

    if rank == 0:
        if e2e is not None:
            out["c5_end_to_end"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, m, args.cpu_sample_points, nvec if c5 else 1)
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist and group_ok:
        dist.barrier()
        dist.destroy_process_group()
    # exit codes, all AFTER the line: 4 = the scatter / compute / gather leg failed, 5 = the timed step's output is wrong
    code = final_exit_code(group_ok, check if rank == 0 else None)
    if code == EXIT_E2E:
        sys.stdout.flush()
        if rank != 0:
            time.sleep(2)  # (rank 0's line first)
        os._exit(code)  # (not sys.exit: a broken process group can hang interpreter shutdown in its destructor)
    if code:
        sys.stderr.write("bench.py: self-check FAILED: %s\n" % json.dumps(check))
        sys.exit(code)


def guarded_leg(leg, timeout_s, rank, out):
    """Runs the multi-GPU leg under a watchdog.  Returns (its result, True), or ({"error": ...}, False) when it raised (a
    failed collective; a verification MISMATCH is a SystemExit(3) and passes through).  A leg that HANGS is given up after
    `timeout_s`: rank 0 prints the line -- every other figure in `out` was final before the leg started -- with the error in
    the leg's place, and EVERY rank leaves with exit code 4, so that the launcher, torchrun and the driver see that the
    path's one exchange did not complete (round 5 exited 0 here).  The process ends itself; nothing is re-executed."""
    import threading

    def timed_out():
        if rank == 0:
            out["c5_end_to_end"] = {"error": "no result after %.0f s (--e2e-timeout): the scatter / compute / gather leg hung; every other "
                                             "figure of this line was final before the leg started" % timeout_s}
            print(json.dumps(out))
            sys.stdout.flush()
        else:
            time.sleep(5)
        os._exit(EXIT_E2E)

    done = threading.Event()
    threading.Thread(target=lambda: done.wait(timeout_s) or timed_out(), daemon=True).start()
    try:
        return leg(), True
    except Exception as exc:  # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[-400:])}, False
    finally:
        done.set()


def final_exit_code(group_ok, check):
    """0, or -- after the JSON line is out -- 4 when the multi-GPU leg failed, 5 when the self-check of the timed output did."""
    if not group_ok:
        return EXIT_E2E
    if check is not None and not check["ok"]:
        return EXIT_SELF_CHECK
    return 0


XGMI_LINK_GBS = 153.0  # per direction per link; rank 0 has one link to each of its 7 peers (SURVEY.md 5 / 8e, MI355X guide)


def link_model(world, vectors_per_gpu, points, chunk_vectors, elem_bytes=8):
    """What the chunked scatter / compute / gather should cost if only the wires counted (xGMI is point to
    point, every peer's shard travels on its own link, 153 GB/s per direction): the pipeline runs ceil(per / chunk) + 2
    lock-step rounds (basic_dsp_amd/batch.py), each moving at most one chunk up and one chunk down per link, full duplex.
    Rank 0's HBM serves all links at once: (world - 1) chunks read and written per round.  `ms` well above `expected_ms`
    on the first N > 1 record means launch / rendezvous latency per round, not bandwidth; `bound` names the larger term."""
    if world <= 1:
        return {"expected_ms": None, "note": "world size 1: no peer, nothing travels"}
    chunks = -(-vectors_per_gpu // chunk_vectors)
    chunk_bytes = chunk_vectors * points * elem_bytes
    round_link_ms = chunk_bytes / (XGMI_LINK_GBS * 1e9) * 1e3
    round_hbm_ms = 2 * (world - 1) * chunk_bytes / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3
    rounds = chunks + 2
    link_ms, hbm_ms = rounds * round_link_ms, rounds * round_hbm_ms
    return {"expected_ms": max(link_ms, hbm_ms), "bound": "xgmi link" if link_ms >= hbm_ms else "rank 0 HBM",
            "link_ms": link_ms, "rank0_hbm_ms": hbm_ms, "rounds": rounds, "bytes_per_peer_per_direction": vectors_per_gpu * points * elem_bytes,
            "link_GBs_per_direction": XGMI_LINK_GBS, "hbm_GBs": HBM_ACHIEVABLE_GBS,
            "not_modelled": "compute (overlapped on a side stream), per-round launch / rendezvous latency"}


def c5_end_to_end(args, bd, torch, dist, dev, rank, world, n, m, nvec, use_dist):
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
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        out = scatter_process_gather_chunked(batch, taps, n, process_shard_gpu, chunk_vectors=args.chunk_vectors, device=dev)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        times.append(time.perf_counter() - t0)
        del out
    best = min(times[1:])
    model = link_model(world, nvec, n, args.chunk_vectors)
    return {"vectors": total, "chunk_vectors": args.chunk_vectors, "ms": best * 1e3, "expected_ms": model["expected_ms"], "link_model": model,
            "Msamples_s": total * n / best / 1e6, "runs_ms": [t * 1e3 for t in times]}


def verified_scatter_gather(args, torch, dist, dev, rank, world, m, share_gpu, compute=None, points=None, corrupt=False):
    """The path's multi-GPU exchange, run AND verified from the headline mode whenever the ranks form a process group
    (`--gpus N` with N > 1, or `--init-dist`): a C5-shaped batch of `--e2e-vectors-per-gpu` vectors of 2^20 points per
    rank lives in rank 0's HBM, goes out in chunks of `--e2e-chunk-vectors` over grouped point-to-point sends (RCCL over
    xGMI), every rank runs convolve_signal -> plain_fft on its chunks while the next ones are in flight, the spectra
    come back two rounds behind (basic_dsp_amd.batch.scatter_process_gather_chunked; replaces the matrix crate's row
    loop, matrix/src/lib.rs:195-208).  One untimed run, three timed ones.  Then rank 0 recomputes the first and the last
    chunk of every peer itself and compares the gathered rows BIT FOR BIT (same kernels, same chunk shape, same data ->
    the same bits on every GPU); at world size 1 it does so for its own shard.  A mismatch ends every rank non-zero.
    Under the one-GPU test hook the ranks talk gloo, so the chunks travel as host tensors and the compute step copies
    them to GPU 0 and back.
    `compute` / `points` / `corrupt`: for the CPU tier of the tests only (tests/test_batch_gloo.py) -- a stand-in for the
    compute step on host tensors, so that the leg's control logic (which rows are verified, the exit on a mismatch) runs at
    world sizes 2 and 3 over gloo without a GPU; the product path never passes them."""
    from basic_dsp_amd.batch import scatter_process_gather_chunked, shard_bounds
    if compute is None:
        from basic_dsp_amd.batch import process_shard_gpu
    else:
        process_shard_gpu = compute
    sync = torch.cuda.synchronize if torch.device(dev).type == "cuda" else (lambda: None)
    n, per, chunk = points or C5_POINTS, args.e2e_vectors_per_gpu, args.e2e_chunk_vectors
    total = per * world
    comm_dev = torch.device("cpu") if share_gpu else dev
    batch = taps = None
    if rank == 0:
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        batch = torch.rand((total, 2 * n), generator=g, device=dev, dtype=torch.float32) * 20 - 10
        taps = (torch.rand(2 * m, generator=g, device=dev, dtype=torch.float32) * 2 - 1) / m
    if share_gpu:
        def fn(shard, tp, points):
            return process_shard_gpu(shard.to(dev), tp.to(dev), points).cpu()
        cbatch, ctaps = (batch.cpu(), taps.cpu()) if rank == 0 else (None, None)
    else:
        fn, cbatch, ctaps = process_shard_gpu, batch, taps
    if share_gpu and os.environ.get("BDSP_BENCH_HANG_E2E") == "1" and rank == world - 1:
        time.sleep(3600)  # TEST HOOK (with --test-share-gpu only): a rank that never joins the leg -- the watchdog must end the run
    times, out = [], None
    for it in range(4):
        out = None
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        out = scatter_process_gather_chunked(cbatch, ctaps, n, fn, chunk_vectors=chunk, device=comm_dev)
        sync()
        dist.barrier()
        times.append(time.perf_counter() - t0)
    ok, rows, bad = 1, 0, []
    if rank == 0 and (corrupt or (share_gpu and os.environ.get("BDSP_BENCH_CORRUPT_E2E") == "1")):
        out[total - 1, min(12345, 2 * n - 1)] += 1.0  # TEST HOOK (with --test-share-gpu only): the verification below must catch this
    if rank == 0:
        for peer in (range(1, world) if world > 1 else [0]):
            f, l = shard_bounds(total, world, peer)
            spans = [(a, min(a + chunk, l)) for a in range(f, l, chunk)]
            for a, b in sorted({spans[0], spans[-1]}):
                ref = process_shard_gpu(batch[a:b], taps, n)
                got = out[a:b].to(dev)
                for v in range(a, b):
                    rows += 1
                    if not torch.equal(got[v - a], ref[v - a]):
                        ok = 0
                        bad.append(v)
    flag = torch.tensor([ok], device=torch.device("cpu") if share_gpu else dev, dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if rank == 0:
            sys.stderr.write("bench.py: gathered rows %s differ from rank 0's own computation of the same rows\n" % bad[:8])
        dist.barrier()
        dist.destroy_process_group()
        raise SystemExit(3)
    best = min(times[1:])
    model = link_model(world, per, n, chunk)
    return {"ms": best * 1e3, "expected_ms": model["expected_ms"], "link_model": model,
            "Msamples_s": total * n / best / 1e6, "verified_rows": rows, "peers": world - 1,
            "verified": "bit-identical to rank 0's own convolve_signal -> plain_fft of the same chunks" +
                        (" (world size 1: rank 0's own shard)" if world == 1 else ""),
            "vectors": total, "points": n, "vectors_per_gpu": per, "chunk_vectors": chunk, "taps": m,
            "runs_ms": [t * 1e3 for t in times], "transport": "gloo, host tensors (one-GPU test hook)" if share_gpu else "RCCL (backend nccl), device tensors"}


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
