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
                ops.append(dist.P2POp(dist.isend, full[a:b].contiguous(), peer, group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return full[start:stop]
    local = torch.empty((stop - start,) + tuple(item_shape), dtype=dtype, device=device)
    if stop > start:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, local, src, group)]):
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
                ops.append(dist.P2POp(dist.irecv, full[a:b], peer, group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return full
    if local.shape[0] > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst, group)]):
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
