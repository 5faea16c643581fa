"""Multi-process test of the batch driver's sharding / scatter / gather logic (BASELINE config C5)
on CPU: gloo backend, world_size 2 (and 3, to cover a ragged last shard).  The compute step is
injected: here it is the CPU oracle (test infrastructure), on the GPU box it is
basic_dsp_amd.batch.process_shard_gpu."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_process(shard, taps, points):
    import oracle_lib as orc
    out = torch.empty_like(shard)
    h = taps.numpy()
    for i in range(shard.shape[0]):
        code, y, _ = orc.convolve_signal(shard[i].numpy(), h, True)
        assert code == 0
        out[i] = torch.from_numpy(orc.fft(y))
    return out


def _worker(rank, world, port, nvec, points, m, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basic_dsp_amd.batch import scatter_process_gather, shard_bounds
    import oracle_lib as orc
    batch = taps = None
    if rank == 0:
        batch = torch.from_numpy(np.stack([orc.fill_uniform(2 * points, 201511212 + v, -10, 10, np.float64)
                                           for v in range(nvec)]))
        taps = torch.from_numpy(orc.fill_uniform(2 * m, 201601172, -1, 1, np.float64))
    seen = []

    def process(shard, t, p):
        seen.append(shard.shape[0])
        return _oracle_process(shard, t, p)

    out = scatter_process_gather(batch, taps, points, process, device=torch.device("cpu"))
    f, l = shard_bounds(nvec, world, rank)
    assert seen == ([l - f] if l > f else [])
    if rank == 0:
        ref = _oracle_process(batch, taps, points)
        err = float((out - ref).abs().max())
        with open(result_path, "w") as fh:
            fh.write("%g" % err)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nvec", [(2, 8), (3, 7), (2, 1)])
def test_scatter_process_gather_gloo(tmp_path, world, nvec):
    port = _free_port()
    result = str(tmp_path / "err.txt")
    mp.spawn(_worker, args=(world, port, nvec, 300, 9, result), nprocs=world, join=True)
    assert float(open(result).read()) == 0.0


def _worker_chunked(rank, world, port, nvec, points, m, chunk, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basic_dsp_amd.batch import scatter_process_gather_chunked, shard_bounds
    import oracle_lib as orc
    batch = taps = None
    if rank == 0:
        batch = torch.from_numpy(np.stack([orc.fill_uniform(2 * points, 201511212 + v, -10, 10, np.float64)
                                           for v in range(nvec)]))
        taps = torch.from_numpy(orc.fill_uniform(2 * m, 201601172, -1, 1, np.float64))
    seen = []

    def process(shard, t, p):
        seen.append(shard.shape[0])
        return _oracle_process(shard, t, p)

    out = scatter_process_gather_chunked(batch, taps, points, process, chunk_vectors=chunk, device=torch.device("cpu"))
    f, l = shard_bounds(nvec, world, rank)
    # every vector of the shard went through the compute step exactly once, in chunks of at most `chunk`
    assert sum(seen) == l - f and all(0 < c <= chunk for c in seen)
    assert len(seen) == -(-(l - f) // chunk)
    if rank == 0:
        ref = _oracle_process(batch, taps, points)
        err = float((out - ref).abs().max())
        with open(result_path, "w") as fh:
            fh.write("%g" % err)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nvec,chunk", [(2, 8, 2), (2, 8, 3), (3, 7, 1), (2, 1, 4), (3, 20, 4)])
def test_scatter_process_gather_chunked_gloo(tmp_path, world, nvec, chunk):
    """The pipelined scatter / compute / gather (chunk k+1 in flight while chunk k is transformed, results two
    rounds behind): bit-identical to transforming the whole batch on rank 0."""
    port = _free_port()
    result = str(tmp_path / "err.txt")
    mp.spawn(_worker_chunked, args=(world, port, nvec, 300, 9, chunk, result), nprocs=world, join=True)
    assert float(open(result).read()) == 0.0


def _fast_process(shard, taps, points):
    """A cheap stand-in for the compute step with the same data dependencies (every output row depends on every
    sample of its input row and on the taps): exact integer-valued arithmetic in f64, so results are bit-identical
    however the rows are grouped."""
    w = torch.arange(1, shard.shape[1] + 1, dtype=shard.dtype)
    return torch.flip(shard, dims=[1]) * 2.0 + (shard * w).sum(dim=1, keepdim=True) + taps.sum()


def _worker_c5_shape(rank, world, port, nvec, points, chunk, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basic_dsp_amd.batch import scatter_process_gather, scatter_process_gather_chunked, shard_bounds
    batch = taps = keep = None
    if rank == 0:
        g = torch.Generator().manual_seed(512)
        batch = torch.randint(-8, 9, (nvec, 2 * points), generator=g).to(torch.float64)
        taps = torch.randint(-3, 4, (2 * 9,), generator=g).to(torch.float64)
        keep = batch.clone()
    f, l = shard_bounds(nvec, world, rank)
    errs = []
    for driver in ("chunked", "whole"):
        seen = []

        def process(shard, t, p):
            seen.append(shard.shape[0])
            return _fast_process(shard, t, p)
        if driver == "chunked":
            out = scatter_process_gather_chunked(batch, taps, points, process, chunk_vectors=chunk, device=torch.device("cpu"))
            # 64 vectors per rank in chunks of 8: eight compute calls of 8 vectors on EVERY rank (8 rounds + 2 of lag)
            assert seen == [chunk] * ((l - f) // chunk), (rank, seen)
        else:
            out = scatter_process_gather(batch, taps, points, process, device=torch.device("cpu"))
            assert seen == [l - f], (rank, seen)
        assert (f, l) == (rank * (nvec // world), (rank + 1) * (nvec // world))  # v -> rank v // 64
        if rank == 0:
            assert torch.equal(batch, keep), "the drivers only read the caller's batch"
            errs.append(float((out - _fast_process(keep, taps, points)).abs().max()))
        else:
            assert out is None
    if rank == 0:
        with open(result_path, "w") as fh:
            fh.write(" ".join("%g" % e for e in errs))
    dist.barrier()
    dist.destroy_process_group()


def test_c5_shape_at_world_size_8_gloo(tmp_path):
    """BASELINE config C5's SHAPE without hardware: 8 ranks, 512 vectors, 64 per rank, chunks of 8 vectors = 8 rounds
    + 2 of lag with 7 peers per round, for both drivers (the 1 048 576-point vectors shrunk to 256 points: the sharding,
    the round structure and the message pattern do not depend on the length).  Asserts the 64-per-rank map, the number
    and size of the compute calls on every rank, bit-identity with the single-process result, and that rank 0's batch
    is left untouched.  Counterpart of the reference's sequential row loop, matrix/src/lib.rs:195-208."""
    port = _free_port()
    result = str(tmp_path / "err.txt")
    mp.spawn(_worker_c5_shape, args=(8, port, 512, 256, 8, result), nprocs=8, join=True)
    assert [float(x) for x in open(result).read().split()] == [0.0, 0.0]


def test_shard_bounds_contiguous_cover():
    from basic_dsp_amd.batch import shard_bounds
    for nvec in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(nvec, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nvec
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
    assert shard_bounds(512, 8, 3) == (192, 256)  # 64 vectors per GPU, contiguous (config C5)


def _bench_leg_worker(rank, world, port, corrupt, result_path):
    """bench.py's verified scatter / compute / gather leg (verified_scatter_gather) with a host stand-in for the compute step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import argparse
    import bench
    calls = []

    def compute(shard, taps, points):  # deterministic, row-wise, depends on the taps: what a mix-up of rows or chunks would break
        calls.append(int(shard.shape[0]))
        return shard * 2.0 + taps[:1] + torch.arange(shard.shape[1], dtype=shard.dtype) * 1e-3

    args = argparse.Namespace(e2e_vectors_per_gpu=5, e2e_chunk_vectors=2)
    try:
        res = bench.verified_scatter_gather(args, torch, dist, torch.device("cpu"), rank, world, 16, True, compute=compute, points=64, corrupt=corrupt)
        code = 0
    except SystemExit as e:
        res, code = None, e.code
    with open(result_path + ".%d" % rank, "w") as fh:
        import json
        json.dump({"code": code, "res": res, "calls": calls}, fh)
    if dist.is_initialized():
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_verified_scatter_gather_leg_over_gloo(world, tmp_path):
    """The leg `bench.py --gpus N` runs after its timed region (round 5), at world sizes 2, 3 and 8 (the node's) on CPU: 5 vectors per rank in
    chunks of 2 (a ragged last chunk), four passes of the chunked pipeline, then rank 0 recomputes the FIRST and the LAST
    chunk of every peer and compares -- 3 rows per peer (2 + 1) -- and a corrupted gathered row ends EVERY rank with exit
    code 3."""
    import json
    for corrupt in (False, True):
        path = str(tmp_path / ("leg%d" % corrupt))
        mp.spawn(_bench_leg_worker, args=(world, _free_port(), corrupt, path), nprocs=world, join=True)
        rs = [json.load(open(path + ".%d" % r)) for r in range(world)]
        if corrupt:
            assert all(r["code"] == 3 for r in rs), rs
            continue
        assert all(r["code"] == 0 for r in rs)
        e = rs[0]["res"]
        assert e["peers"] == world - 1 and e["verified_rows"] == 3 * (world - 1) and e["vectors"] == 5 * world
        assert e["chunk_vectors"] == 2 and e["ms"] > 0 and len(e["runs_ms"]) == 4
        # every rank ran its 3 chunks (2 + 2 + 1 vectors) in each of the four passes; rank 0 also the 2 verification chunks per peer
        for r in range(1, world):
            assert rs[r]["calls"] == [2, 2, 1] * 4, rs[r]["calls"]
        assert rs[0]["calls"] == [2, 2, 1] * 4 + [2, 1] * (world - 1), rs[0]["calls"]


_GUARDED_LEG_RANK = r"""
import json, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import argparse, bench
rank, world, mode = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
out = {"value": 123.0, "value_by_rank": [1.0] * world}  # "every other figure of the line was final before the leg started"

def compute(shard, taps, points):
    return shard * 2.0 + taps[:1]

def leg():
    if mode == "hang" and rank == world - 1:
        time.sleep(3600)       # a rank that never joins the leg
    if mode == "raise" and rank == world - 1:
        raise RuntimeError("collective failed on this rank")
    args = argparse.Namespace(e2e_vectors_per_gpu=5, e2e_chunk_vectors=2)
    return bench.verified_scatter_gather(args, torch, dist, torch.device("cpu"), rank, world, 16, True, compute=compute, points=64)

e2e, ok = bench.guarded_leg(leg, 6.0 if mode == "hang" else 60.0, rank, out)
if rank == 0:
    out["c5_end_to_end"] = e2e
    print(json.dumps(out), flush=True)
code = bench.final_exit_code(ok, None)
if code:
    os._exit(code)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3, 8])
def test_a_hung_or_failed_leg_ends_with_exit_code_4_after_the_line(world, tmp_path):
    """Round 6 (VERDICT r5 item 4a / ADVICE): bench.py's watchdog around the scatter / compute / gather leg.  A leg that hangs
    (the last rank never joins it) is given up after the timeout: rank 0 prints the line, whose other figures were final,
    with an error in the leg's place, and EVERY rank leaves with exit code 4 -- no rank exits 0, nothing is re-executed.
    A leg that completes carries the link model's `expected_ms` next to `ms`.  bench.supervise (the launcher's side) keeps
    waiting for rank 0's line when a PEER leaves with code 4 first."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    def run(mode, supervised):
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, "-c", _GUARDED_LEG_RANK, root, mode], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        if supervised:
            failed, out0 = bench.supervise(procs, 120.0)
        else:  # every rank is left to end by itself
            out0 = procs[0].communicate(timeout=120)[0]
            for p in procs[1:]:
                p.wait(timeout=120)
            failed = None
        lines = [l for l in out0.decode().splitlines() if l.startswith("{")]
        return failed, [p.returncode for p in procs], [json.loads(l) for l in lines]

    failed, codes, lines = run("ok", True)
    assert failed is None and codes == [0] * world and len(lines) == 1
    e = lines[0]["c5_end_to_end"]
    assert e["verified_rows"] == 3 * (world - 1) and e["ms"] > 0
    # 5 vectors of 64 points in chunks of 2: 3 + 2 lock-step rounds of 2 * 64 * 8 bytes per link at 153 GB/s
    assert abs(e["expected_ms"] - 5 * (2 * 64 * 8) / 153e9 * 1e3) < 1e-12 and e["link_model"]["rounds"] == 5 and e["link_model"]["bound"] == "xgmi link"
    # a hang: every rank ends ITSELF with code 4, rank 0 after printing the line
    failed, codes, lines = run("hang", False)
    assert codes == [4] * world, codes
    assert len(lines) == 1 and "hung" in lines[0]["c5_end_to_end"]["error"] and lines[0]["value"] == 123.0
    # a collective that fails on a peer: the peer leaves with code 4 at once, the launcher's supervisor keeps waiting for
    # rank 0, whose own collective then fails -- its line still arrives, with the error in the leg's place, and it exits 4 too
    failed, codes, lines = run("raise", True)
    assert failed and "code 4" in failed and codes[0] == 4 and codes[world - 1] == 4 and all(c != 0 for c in codes), (failed, codes)
    assert len(lines) == 1 and "error" in lines[0]["c5_end_to_end"] and lines[0]["value"] == 123.0


def test_exit_codes_and_link_model_of_the_bench_line():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    assert bench.final_exit_code(True, None) == 0 and bench.final_exit_code(True, {"ok": True}) == 0
    assert bench.final_exit_code(True, {"ok": False}) == bench.EXIT_SELF_CHECK == 5
    assert bench.final_exit_code(False, {"ok": False}) == bench.EXIT_E2E == 4
    # config C5 on the node: 64 vectors of 2^20 points per GPU in chunks of 8 -> 8 + 2 rounds of 64 MiB per link
    m = bench.link_model(8, 64, 1 << 20, 8)
    assert m["rounds"] == 10 and abs(m["expected_ms"] - 10 * (8 << 23) / 153e9 * 1e3) < 1e-9 and m["bound"] == "xgmi link"
    assert m["bytes_per_peer_per_direction"] == 512 << 20 and 0 < m["rank0_hbm_ms"] < m["link_ms"]
    assert bench.link_model(1, 8, 1 << 20, 2)["expected_ms"] is None
    # the self-check's DFT-by-definition helper against numpy's transform
    import numpy as np
    rng = np.random.default_rng(5)
    for n in (1 << 12, 3000, 1 << 15):
        y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        ks = sorted({0, 1, n // 2, n - 1, 12345678 % n})
        assert np.max(np.abs(bench.dft_bins(y, ks) - np.fft.fft(y)[ks])) < 1e-9 * np.sqrt(n)
