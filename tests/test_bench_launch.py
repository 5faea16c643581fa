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


def test_cpu_baseline_fields_are_numeric():
    sys.path.insert(0, ROOT)
    import bench
    b = bench.cpu_baseline(1 << 15, 64, 1 << 15)
    assert b["kind"] == "port" and b["cores"] >= 1 and b["unit"] == "Msamples/s"
    for k in ("value", "reference_schedule_1core_Msamples_s", "fair_1core_Msamples_s", "fair_allcores_Msamples_s"):
        assert isinstance(b[k], float) and b[k] > 0
