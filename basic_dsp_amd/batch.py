"""Batch driver: shard independent vectors across the GPUs of one node (BASELINE config C5).

The reference has no distributed layer; its only batching is the matrix crate's sequential loop over
rows (matrix/src/lib.rs:195-208).  Vectors are independent, so the path shards by vector with no
collective inside an operation:

    rank 0 holds the batch  --scatter-->  every rank runs the fused kernels on its shard
                            <--gather---

One process per GPU (torchrun); scatter/gather use torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU unit tests).  Vector v goes to rank v // ceil(V / world):
contiguous blocks, so each peer's shard is ONE contiguous message on its own xGMI link.

The compute step (`process_shard_gpu`) calls the C ABI of libbasic_dsp_hip.so and needs a GPU; the
sharding logic takes the compute step as a parameter so it can be exercised without one.
"""
import ctypes as C

import torch
import torch.distributed as dist


def shard_bounds(num_vectors, world_size, rank):
    """[first, last) of the contiguous block of vectors owned by `rank`."""
    per = -(-num_vectors // world_size)
    first = min(rank * per, num_vectors)
    return first, min(first + per, num_vectors)


def process_shard_gpu(shard, taps, points, stream=None):
    """convolve_signal(shared taps) then plain_fft on every vector of `shard`.

    shard: [n_vec, 2*points] f32/f64 CUDA tensor (interleaved complex); taps: [2*M] CUDA tensor.
    Returns a tensor of the same shape holding the spectra.  Two batched launches for the
    convolution (spectrum of the taps + fused overlap-save over all vectors) and the batched FFT.
    """
    from . import _lib
    lib = _lib.lib
    assert shard.is_cuda and shard.is_contiguous()
    elem = 0 if shard.dtype == torch.float32 else 1
    nvec = shard.shape[0]
    m = taps.numel() // 2
    sp = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
    spec = torch.empty(2 * lib.bdsp_hip_conv_spectrum_points(), device=shard.device, dtype=shard.dtype)
    out = torch.empty_like(shard)
    _lib.check(lib.bdsp_hip_dev_conv_prepare(elem, taps.data_ptr(), m, spec.data_ptr(), sp), "conv_prepare")
    _lib.check(lib.bdsp_hip_dev_convolve_prepared(elem, shard.data_ptr(), out.data_ptr(), points, nvec,
                                                  spec.data_ptr(), m, sp), "convolve")
    flag = C.c_int(0)
    _lib.check(lib.bdsp_hip_dev_fft(elem, out.data_ptr(), shard.data_ptr(), points, nvec, 0, 1.0, -1, 0.0,
                                    C.byref(flag), sp), "fft")
    return shard if flag.value else out


def scatter_process_gather(batch, taps, points, process_fn, group=None, device=None):
    """Scatter `batch` ([V, 2*points], meaningful on rank 0 only) from rank 0, run
    process_fn(shard, taps, points) on every rank, gather the results back to rank 0.

    Point-to-point batched sends (one message per peer, all in flight together) rather than a ring:
    xGMI is point-to-point, so each of the 7 links carries exactly one peer's shard.
    Returns the gathered [V, 2*points] tensor on rank 0, None elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = [None]
    if rank == 0:
        meta = [(batch.shape[0], str(batch.dtype).split(".")[-1])]
    dist.broadcast_object_list(meta, src=0, group=group)
    nvec, dtype_name = meta[0]
    dtype = getattr(torch, dtype_name)
    if device is None:
        device = batch.device if rank == 0 else torch.device("cpu")
    first, last = shard_bounds(nvec, world, rank)
    if rank == 0:
        shard = batch[first:last].contiguous()
        ops = []
        for peer in range(1, world):
            f, l = shard_bounds(nvec, world, peer)
            if l > f:
                ops.append(dist.P2POp(dist.isend, batch[f:l].contiguous(), peer, group))
    else:
        shard = torch.empty((last - first, 2 * points), dtype=dtype, device=device)
        ops = [dist.P2POp(dist.irecv, shard, 0, group)] if last > first else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    # the shared filter goes to everybody (8 KiB: one broadcast)
    taps_meta = [taps.numel() if rank == 0 else None]
    dist.broadcast_object_list(taps_meta, src=0, group=group)
    if rank != 0:
        taps = torch.empty(taps_meta[0], dtype=dtype, device=device)
    dist.broadcast(taps, src=0, group=group)

    result = process_fn(shard, taps, points) if last > first else shard

    if rank == 0:
        out = torch.empty((nvec, 2 * points), dtype=dtype, device=device)
        out[first:last] = result
        ops = []
        for peer in range(1, world):
            f, l = shard_bounds(nvec, world, peer)
            if l > f:
                ops.append(dist.P2POp(dist.irecv, out[f:l], peer, group))
    else:
        out = None
        ops = [dist.P2POp(dist.isend, result.contiguous(), 0, group)] if last > first else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out
