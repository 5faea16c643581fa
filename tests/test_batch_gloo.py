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


def test_shard_bounds_contiguous_cover():
    from basic_dsp_amd.batch import shard_bounds
    for nvec in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(nvec, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nvec
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
    assert shard_bounds(512, 8, 3) == (192, 256)  # 64 vectors per GPU, contiguous (config C5)
