"""Batch sharding of independent stereo pairs across the GPUs of one node.

The reference has no distributed layer (its only parallelism is cv::parallel_for_ stripes inside one
image: FGS.cpp:235-243, DF.cpp:158); stereo pairs are independent units, so the batch shards
embarrassingly: rank g owns the contiguous pairs [g*N/G, (g+1)*N/G).  The only exchange is the batch
scatter (inputs, root -> ranks) and gather (filtered maps, ranks -> root), done with point-to-point
send/recv over torch.distributed -- backend "nccl" is RCCL on ROCm, and on an MI355X node every peer
is one direct xGMI link away, so root->7 peers proceeds on seven links at once (a ring collective
would be bound by one link).  No collective touches the filter itself.

One process per GPU; works unchanged on the "gloo" backend with CPU tensors (tests/).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous shard [start, stop) of rank `rank`; the first n_total % world ranks get one extra."""
    base, extra = divmod(int(n_total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _wire(t):
    """The tensor as the process group sees it.  RCCL's torch binding moves 1-, 4- and 8-byte integers, halves, floats and
    doubles but REFUSES 16-bit integers ("Input tensor data type is not supported for NCCL process group: Short",
    found by tests/test_gpu_rccl_one_rank.py on the one-GPU box) -- and the disparity maps are CV_16S.  A contiguous
    int16 / uint16 block therefore travels as its bytes: the same memory, no copy, the receiver's view writes straight
    into the int16 tensor."""
    if t.dtype in (torch.int16, torch.uint16):
        if not t.is_contiguous():
            raise ValueError("a 16-bit block must be contiguous to travel as bytes")
        return t.view(torch.uint8)
    return t


def scatter_batch(full, n_total, item_shape, dtype, device, src=0, group=None):
    """Root holds `full` (n_total, *item_shape); every rank returns its own shard.

    Non-root ranks pass full=None.  Point-to-point: the root posts one isend per peer, each peer one
    irecv; the root's own shard is a view (no copy).
    """
    rank, world = _world(group)
    start, stop = shard_range(n_total, rank, world)
    if world == 1:
        return full[start:stop]
    if rank == src:
        ops = []
        for peer in range(world):
            if peer == src:
                continue
            a, b = shard_range(n_total, peer, world)
            if b > a:
                ops.append(dist.P2POp(dist.isend, _wire(full[a:b].contiguous()), peer, group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return full[start:stop]
    local = torch.empty((stop - start,) + tuple(item_shape), dtype=dtype, device=device)
    if stop > start:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, _wire(local), src, group)]):
            w.wait()
    return local


def gather_batch(local, n_total, dst=0, group=None):
    """Inverse of scatter_batch: the root returns (n_total, *item_shape), the others None."""
    rank, world = _world(group)
    if world == 1:
        return local
    if rank == dst:
        full = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        start, stop = shard_range(n_total, rank, world)
        full[start:stop] = local
        ops = []
        for peer in range(world):
            if peer == dst:
                continue
            a, b = shard_range(n_total, peer, world)
            if b > a:
                ops.append(dist.P2POp(dist.irecv, _wire(full[a:b]), peer, group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return full
    if local.shape[0] > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, _wire(local.contiguous()), dst, group)]):
            w.wait()
    return None


def max_over_ranks(value, device, group=None):
    """Max of a python float over ranks (timing): the one tiny all-reduce of the harness."""
    rank, world = _world(group)
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value, device, group=None):
    rank, world = _world(group)
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())


def min_over_ranks(value, device, group=None):
    """Min of a python float over ranks (all ranks agree on a pass / fail flag)."""
    rank, world = _world(group)
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item())


def gather_scalars(value, device, group=None):
    """[value of rank 0, value of rank 1, ...] on every rank: one all-reduce of a world-sized vector."""
    rank, world = _world(group)
    if world == 1:
        return [float(value)]
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[rank] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(v) for v in t.tolist()]


def gather_objects(obj, group=None):
    """[obj of rank 0, ...] on every rank (small python objects: check reports)."""
    rank, world = _world(group)
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj, group=group)
    return out


def sub_range(n_total, rank, world, sub, n_sub):
    """Global [start, stop) of sub-batch `sub` of `n_sub` inside the shard of `rank`."""
    a, b = shard_range(n_total, rank, world)
    lo, hi = shard_range(b - a, sub, n_sub)
    return a + lo, a + hi


def pipelined_scatter_filter_gather(full, n_total, in_shapes, in_dtypes, out_shape, out_dtype, device, process,
                                    n_sub=4, src=0, group=None):
    """Serving-shaped flow of SURVEY.md 8(e): the root holds the whole batch (`full` = list of input tensors of
    shape (n_total, *in_shapes[i]); None elsewhere); every rank's shard is cut into `n_sub` sub-batches and the
    transfers of sub-batch s+1 (root -> ranks) and s-1 (ranks -> root) are in flight while sub-batch s is filtered.

    `process(*inputs_of_the_sub_batch, out_of_the_sub_batch)` filters one sub-batch (asynchronously on the
    current stream for device tensors).  Point-to-point only: on an MI355X node every peer is one xGMI link
    from the root, so the root's seven sends proceed on seven links at once.  Ordering on CUDA: torch's NCCL
    work is enqueued behind what the current stream holds at that moment and `wait()` makes the current stream
    wait for that one transfer, so issue order alone gives  scatter(s+1) || filter(s) || gather(s-1).

    Returns (stats, full_out): stats = dict(total_ms, sub_batches) on every rank (max over ranks, barrier to
    barrier), full_out = (n_total, *out_shape) on the root, None elsewhere."""
    import time

    rank, world = _world(group)
    a, b = shard_range(n_total, rank, world)
    n_local = b - a
    n_sub = max(1, min(int(n_sub), max(1, n_total // max(world, 1))))      # the same on every rank
    cuda = torch.device(device).type == "cuda"

    def dsync():
        if cuda:
            torch.cuda.synchronize(device)

    if rank == src:
        local_in = [t[a:b] for t in full]
        full_out = torch.empty((n_total,) + tuple(out_shape), dtype=out_dtype, device=device)
        local_out = full_out[a:b]
    else:
        local_in = [torch.empty((n_local,) + tuple(s), dtype=d, device=device) for s, d in zip(in_shapes, in_dtypes)]
        full_out = None
        local_out = torch.empty((n_local,) + tuple(out_shape), dtype=out_dtype, device=device)

    def post(ops):
        return dist.batch_isend_irecv(ops) if ops else []

    def post_scatter(s):
        ops = []
        if world == 1:
            return []
        if rank == src:
            for peer in range(world):
                if peer == src:
                    continue
                lo, hi = sub_range(n_total, peer, world, s, n_sub)
                if hi > lo:
                    ops += [dist.P2POp(dist.isend, _wire(t[lo:hi]), peer, group) for t in full]
        else:
            lo, hi = sub_range(n_total, rank, world, s, n_sub)
            if hi > lo:
                ops += [dist.P2POp(dist.irecv, _wire(t[lo - a:hi - a]), src, group) for t in local_in]
        return post(ops)

    def post_gather(s):
        ops = []
        if world == 1:
            return []
        if rank == src:
            for peer in range(world):
                if peer == src:
                    continue
                lo, hi = sub_range(n_total, peer, world, s, n_sub)
                if hi > lo:
                    ops.append(dist.P2POp(dist.irecv, _wire(full_out[lo:hi]), peer, group))
        else:
            lo, hi = sub_range(n_total, rank, world, s, n_sub)
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, _wire(local_out[lo - a:hi - a]), src, group))
        return post(ops)

    dsync()
    if world > 1:
        dist.barrier(group)
    t0 = time.perf_counter()
    pending = post_scatter(0)
    gathers = []
    for s in range(n_sub):
        for w in pending:
            w.wait()
        pending = post_scatter(s + 1) if s + 1 < n_sub else []
        lo, hi = sub_range(n_total, rank, world, s, n_sub)
        if hi > lo:
            process(*[t[lo - a:hi - a] for t in local_in], local_out[lo - a:hi - a])
        gathers += post_gather(s)
    for w in gathers:
        w.wait()
    dsync()
    if world > 1:
        dist.barrier(group)
    total = max_over_ranks(time.perf_counter() - t0, device if cuda else torch.device("cpu"), group)
    return {"total_ms": round(total * 1e3, 3), "sub_batches": n_sub}, full_out
