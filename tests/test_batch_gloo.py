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
