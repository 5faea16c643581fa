"""bench.py's launch contract, exercised without a GPU: `--gpus N` starts N ranks itself (one child per LOCAL_RANK,
created before anything touches a GPU), refuses a --gpus that disagrees with the launcher's WORLD_SIZE, and the CPU
baseline leg returns numeric fields."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(kw)
    return env


def test_gpus_n_launches_n_children_dry_run():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--dry-run-launch"], env=_env(),
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    launches = [json.loads(l)["launch"] for l in p.stdout.splitlines() if l.strip()]
    assert len(launches) == 2
    for r, l in enumerate(launches):
        assert l["env"]["RANK"] == str(r) and l["env"]["LOCAL_RANK"] == str(r) and l["env"]["WORLD_SIZE"] == "2"
        assert l["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert l["cmd"][1] == BENCH and "--dry-run-launch" not in l["cmd"] and "--gpus" in l["cmd"]
    assert launches[0]["env"]["MASTER_PORT"] == launches[1]["env"]["MASTER_PORT"]


def test_c5_on_eight_gpus_dry_run_launch():
    """`bench.py --mode c5 --gpus 8`: eight children, RANK = LOCAL_RANK = 0..7, WORLD_SIZE 8, one rendezvous on
    127.0.0.1, every child told --mode c5 --gpus 8 (config C5: 64 vectors per GPU, 512 in all)."""
    p = subprocess.run([sys.executable, BENCH, "--mode", "c5", "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run-launch"],
                       env=_env(), capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    launches = [json.loads(l)["launch"] for l in p.stdout.splitlines() if l.strip()]
    assert len(launches) == 8
    ports = set()
    for r, l in enumerate(launches):
        e = l["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["LOCAL_WORLD_SIZE"]) == (str(r), str(r), "8", "8")
        assert e["MASTER_ADDR"] == "127.0.0.1" and int(e["MASTER_PORT"]) > 0
        ports.add(e["MASTER_PORT"])
        c = l["cmd"]
        assert c[1] == BENCH and "--dry-run-launch" not in c
        assert c[c.index("--mode") + 1] == "c5" and c[c.index("--gpus") + 1] == "8"
    assert len(ports) == 1


def test_gpus_must_match_world_size_under_a_launcher():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_children_that_fail_make_the_parent_fail():
    # no GPU here: every child exits with an error, and so must the parent (never a silent 1-GPU run)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0


def test_a_rank_that_dies_takes_the_others_down_at_once():
    """One rank exits non-zero while the others would sit in a rendezvous for minutes: the launcher's supervisor ends
    them within seconds and reports the failure; and an overall deadline ends ranks that never finish."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    hang = [sys.executable, "-c", "import time; print('partial', flush=True); time.sleep(600)"]
    die = [sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(7)"]
    procs = [subprocess.Popen(hang, stdout=subprocess.PIPE), subprocess.Popen(die), subprocess.Popen(hang, stdout=subprocess.DEVNULL)]
    t0 = time.monotonic()
    failed, out = bench.supervise(procs, 120.0)
    assert failed and "rank 1" in failed and "7" in failed
    assert time.monotonic() - t0 < 20
    assert all(p.returncode is not None for p in procs) and procs[0].returncode != 0
    assert b"partial" in out
    procs = [subprocess.Popen(hang, stdout=subprocess.PIPE), subprocess.Popen(hang, stdout=subprocess.DEVNULL)]
    t0 = time.monotonic()
    failed, _ = bench.supervise(procs, 1.0)
    assert failed and "launch-timeout" in failed and time.monotonic() - t0 < 20
    assert all(p.returncode is not None for p in procs)
    ok = [sys.executable, "-c", "print('{}')"]
    procs = [subprocess.Popen(ok, stdout=subprocess.PIPE), subprocess.Popen(ok, stdout=subprocess.DEVNULL)]
    failed, out = bench.supervise(procs, 60.0)
    assert failed is None and out.strip() == b"{}"


def test_stale_profiles_are_not_quoted(tmp_path, monkeypatch):
    """roofline.traffic / frac_rocprof come from committed rocprofv3 runs; a profile collected on other kernel sources
    than the ones in the tree must yield None, not an old number."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    (tmp_path / "basic_dsp_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "basic_dsp_amd" / "csrc" / "k.hip").write_text("kernel v1")
    sha = bench.kernel_source_sha16()
    tag = bench.PROFILE_TAG
    (tmp_path / "profiles" / (tag + "_profile_meta.json")).write_text(json.dumps({"source_sha16": sha}))
    (tmp_path / "profiles" / (tag + "_hbm_traffic.json")).write_text(json.dumps({"kernels": {
        "k_overlap_save_v2<4, false>": {"hbm_bytes_per_launch": 271e6},
        "k_fft_pass<float, 256, 16, -1, true, false, true>": {"hbm_bytes_per_launch": 268e6},
        "k_fft_pass<float, 256, 16, -1, false, false, true>": {"hbm_bytes_per_launch": 269e6}}}))
    (tmp_path / "profiles" / (tag + "_bench_kernel_stats.csv")).write_text(
        '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
        '"void bdsp::k_overlap_save_v2<4, false>(bdsp::ConvV2Args)",10,600000,60000.0,30,57000,70000,1.0\n'
        '"void bdsp::k_fft_pass<float, 256, 16, -1, true, false, true>(x)",10,450000,45000.0,20,1,2,1.0\n'
        '"void bdsp::k_fft_pass<float, 256, 16, -1, false, false, true>(x)",20,820000,41000.0,40,1,2,1.0\n')
    f = bench.profile_figures()
    assert f["stale"] is False and f["conv_traffic"] == 271e6 and f["fft_traffic"] == 268e6 + 2 * 269e6
    assert f["conv_avg_ns"] == 60000.0 and f["conv_min_ns"] == 57000.0 and f["fft_avg_ns"] == 45000.0 + 2 * 41000.0
    (tmp_path / "basic_dsp_amd" / "csrc" / "k.hip").write_text("kernel v2")
    f = bench.profile_figures()
    assert f["stale"] is True and f["conv_traffic"] is None and f["conv_avg_ns"] is None and f["fft_traffic"] is None


def test_cpu_baseline_fields_are_numeric():
    sys.path.insert(0, ROOT)
    import bench
    b = bench.cpu_baseline(1 << 15, 64, 1 << 15)
    assert b["kind"] == "port" and b["cores"] >= 1 and b["unit"] == "Msamples/s"
    for k in ("value", "reference_schedule_1core_Msamples_s", "fair_1core_Msamples_s", "fair_allcores_Msamples_s"):
        assert isinstance(b[k], float) and b[k] > 0
    # round 6: every leg runs three times (median -> the rates, fastest -> *_best), and a pocketfft / scipy sanity row
    # stands next to the port ("not the reference")
    assert b["reps"] == 3 and all(len(v) == 3 for v in b["seconds_runs"].values())
    for k in ("reference_schedule_1core_Msamples_s", "fair_1core_Msamples_s", "fair_allcores_Msamples_s"):
        assert b[k + "_best"] >= b[k] > 0
    assert b["numpy_sanity_Msamples_s"] > 0 and "NOT the reference" in b["numpy_sanity"]
    assert bench.cpu_baseline(1 << 12, 64, 1 << 12, reps=1)["reps"] == 1
    # --mode c5: whole vectors, one after the other
    b = bench.cpu_baseline(1 << 13, 64, 1 << 15, vectors=64)
    assert "4 whole vectors of 8192 points" in b["sample"] and b["value"] > 0


def test_clock_power_sampler_degrades_to_nulls_without_a_device():
    """bench.py's clock / power figures come from a host thread polling librocm_smi64 during the run; without the library,
    a device or permission every figure is None (the line then says nothing about clocks) -- never an exception and never
    a remembered number."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    s = bench.GpuSampler().start()
    time.sleep(0.02)
    s.stop()
    w = s.window(0.0, 1e18)
    if s.source is None:
        assert w == {"samples": 0, "sclk_mhz": None, "socket_power_w": None} and s.cap_w is None
    else:  # (a box with a GPU: real samples)
        assert w["samples"] >= 1
