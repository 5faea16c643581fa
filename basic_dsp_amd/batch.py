"""Batch driver: shard independent vectors across the GPUs of one node (BASELINE config C5).

The reference has no distributed layer; its only batching is the matrix crate's sequential loop over
rows (matrix/src/lib.rs:195-208).  Vectors are independent, so the path shards by vector with no
collective inside an operation:

    rank 0 holds the batch  --scatter-->  every rank runs the fused kernels on its shard
                            <--gather---

One process per GPU (torchrun); scatter/gather use torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU unit tests).  Vector v goes to rank v // ceil(V / world):
contiguous blocks, so each peer's shard is ONE contiguous message on its own xGMI link.

`scatter_process_gather` moves whole shards (scatter, compute, gather one after the other);
`scatter_process_gather_chunked` cuts every shard into chunks of a few vectors and runs the three
stages as a pipeline: chunk k+1 travels to the peer while chunk k is transformed there and the
result of chunk k-1 travels back (a shard is 512 MiB = ~3.5 ms per
direction per xGMI link, about ten times its compute time).

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
    Returns a NEW tensor of the same shape holding the spectra; `shard` is left untouched (the drivers below hand
    over views of the caller's batch on rank 0).  Two batched launches for the convolution (spectrum of the taps +
    fused overlap-save over all vectors) and the batched FFT, which ping-pongs between the convolution's result and
    a scratch buffer of its own -- round 3 used `shard` as that partner and so overwrote the caller's rows.
    """
    from . import _lib
    lib = _lib.lib
    assert shard.is_cuda and shard.is_contiguous()
    elem = 0 if shard.dtype == torch.float32 else 1
    nvec = shard.shape[0]
    m = taps.numel() // 2
    # torch's default stream has handle 0 = "library stream" in the C ABI: stream_arg() forwards it as HIP's null
    # stream, so the kernels are ordered after the receive that filled `shard` and before whatever reads `out`
    sp = _lib.stream_arg(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
    spec = torch.empty(2 * lib.bdsp_hip_conv_spectrum_points(), device=shard.device, dtype=shard.dtype)
    out = torch.empty_like(shard)
    scratch = torch.empty_like(shard)
    _lib.check(lib.bdsp_hip_dev_conv_prepare(elem, taps.data_ptr(), m, spec.data_ptr(), sp), "conv_prepare")
    _lib.check(lib.bdsp_hip_dev_convolve_prepared(elem, shard.data_ptr(), out.data_ptr(), points, nvec,
                                                  spec.data_ptr(), m, sp), "convolve")
    flag = C.c_int(0)
    _lib.check(lib.bdsp_hip_dev_fft(elem, out.data_ptr(), scratch.data_ptr(), points, nvec, 0, 1.0, -1, 0.0,
                                    C.byref(flag), sp), "fft")
    return scratch if flag.value else out


def scatter_process_gather(batch, taps, points, process_fn, group=None, device=None):
    """Scatter `batch` ([V, 2*points], meaningful on rank 0 only) from rank 0, run
    process_fn(shard, taps, points) on every rank, gather the results back to rank 0.
    `batch` is only read: process_fn must not modify its input (rank 0 hands it views of `batch`).

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


def scatter_process_gather_chunked(batch, taps, points, process_fn, chunk_vectors=8, group=None, device=None):
    """The pipelined form of scatter_process_gather.  Every rank's shard is cut into chunks of `chunk_vectors`
    vectors.  Communication runs in lock-step ROUNDS, one grouped batch of point-to-point operations per round
    and rank (the same content on both sides, so the grouped calls cannot deadlock):

        round r:   rank 0 -> peer : chunk r            peer -> rank 0 : result of chunk r - 2

    and the peer transforms chunk r on a SIDE stream while round r + 1 is in flight (on CUDA; the CPU/gloo path
    of the unit tests computes in line).  The two-round lag of the results keeps a round's issue from waiting for
    the transform that has just been queued.  Rank 0 transforms its own shard chunk by chunk between rounds.
    `batch` is only read: process_fn must not modify its input (rank 0 hands it views of `batch`).
    Returns the gathered [V, 2*points] tensor on rank 0, None elsewhere.
    """
    single = not dist.is_initialized()  # one process, one GPU: the same pipeline without any communication
    world = 1 if single else dist.get_world_size(group)
    rank = 0 if single else dist.get_rank(group)
    meta = [None]
    if rank == 0:
        meta = [(batch.shape[0], str(batch.dtype).split(".")[-1], int(taps.numel()))]
    if not single:
        dist.broadcast_object_list(meta, src=0, group=group)
    nvec, dtype_name, ntaps = meta[0]
    dtype = getattr(torch, dtype_name)
    if device is None:
        device = batch.device if rank == 0 else torch.device("cpu")
    if rank != 0:
        taps = torch.empty(ntaps, dtype=dtype, device=device)
    if not single:
        dist.broadcast(taps, src=0, group=group)
    cuda = torch.device(device).type == "cuda"
    side = torch.cuda.Stream(device=device) if cuda else None

    def chunks_of(r):
        f, l = shard_bounds(nvec, world, r)
        return [(a, min(a + chunk_vectors, l)) for a in range(f, l, chunk_vectors)]

    per_rank = [chunks_of(r) for r in range(world)]
    nrounds = max([len(c) for c in per_rank[1:]] + [0]) + 2  # + the two-round lag of the results
    mine = per_rank[rank]
    out = torch.empty((nvec, 2 * points), dtype=dtype, device=device) if rank == 0 else None
    recv_buf = [torch.empty((b - a, 2 * points), dtype=dtype, device=device) for a, b in mine] if rank != 0 else []
    results = [None] * len(mine)
    done = [None] * len(mine)  # CUDA events: result k is complete on the side stream

    def run_chunk(k, x):
        if cuda:
            main = torch.cuda.current_stream(device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                results[k] = process_fn(x, taps, points)
                done[k] = torch.cuda.Event()
                done[k].record(side)
            # The caching allocator recycles a block for the stream it was allocated on as soon as the tensor dies:
            # tell it about the OTHER stream that touches each buffer.  The result (allocated on the side stream) is
            # sent / copied on the main stream after this function's caller has dropped it; the input (allocated on
            # the main stream) is still being read by the side stream's kernels when the round loop moves on.
            results[k].record_stream(main)
            x.record_stream(side)
        else:
            results[k] = process_fn(x, taps, points)

    own_next = 0
    for r in range(nrounds):
        ops = []
        if rank == 0:
            for peer in range(1, world):
                pc = per_rank[peer]
                if r < len(pc):
                    a, b = pc[r]
                    ops.append(dist.P2POp(dist.isend, batch[a:b], peer, group))
                if 0 <= r - 2 < len(pc):
                    a, b = pc[r - 2]
                    ops.append(dist.P2POp(dist.irecv, out[a:b], peer, group))
        else:
            if r < len(mine):
                ops.append(dist.P2POp(dist.irecv, recv_buf[r], 0, group))
            if 0 <= r - 2 < len(mine):
                if cuda:
                    torch.cuda.current_stream(device).wait_event(done[r - 2])
                ops.append(dist.P2POp(dist.isend, results[r - 2], 0, group))
        works = dist.batch_isend_irecv(ops) if ops else []
        # rank 0 transforms a chunk of its own shard while the round is in flight
        if rank == 0 and own_next < len(mine):
            a, b = mine[own_next]
            run_chunk(own_next, batch[a:b])
            own_next += 1
        for w in works:
            w.wait()
        if rank != 0 and r < len(mine):
            run_chunk(r, recv_buf[r])
    if rank == 0:
        while own_next < len(mine):
            a, b = mine[own_next]
            run_chunk(own_next, batch[a:b])
            own_next += 1
        if cuda:
            torch.cuda.current_stream(device).wait_stream(side)
        for k, (a, b) in enumerate(mine):
            out[a:b] = results[k]
    elif cuda:
        torch.cuda.current_stream(device).wait_stream(side)
    return out
