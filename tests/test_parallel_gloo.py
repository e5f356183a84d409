"""`gloo` tests of the batch scatter/gather harness (the N>1 path of bench.py), on CPU: world size 2, and world size 8 --
BASELINE config 4's (8 ranks of one node: the root's 7-peer point-to-point groups, 8-way ragged shards, root != 0)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from addingdisparityfiltering_amd import parallel


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_16_bit_blocks_travel_as_bytes():
    t = torch.arange(24, dtype=torch.int16).reshape(2, 3, 4)
    w = parallel._wire(t[1:])
    assert w.dtype == torch.uint8 and w.shape == (1, 3, 8) and w.data_ptr() == t[1:].data_ptr()
    w.zero_()
    assert int(t[1:].abs().sum()) == 0 and int(t[0].sum()) == sum(range(12))
    for dt in (torch.uint8, torch.float32, torch.int32):
        x = torch.zeros(3, dtype=dt)
        assert parallel._wire(x) is x
    with pytest.raises(ValueError):
        parallel._wire(t[:, :, ::2])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q, root=0):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        shape = (6, 10)
        # RCCL's binding refuses 16-bit integers (tests/test_gpu_rccl_one_rank.py): nothing may be posted as one
        make_op, posted = dist.P2POp, []
        dist.P2POp = lambda op, tensor, *a, **k: (posted.append(tensor.dtype), make_op(op, tensor, *a, **k))[1]
        full = None
        if rank == root:
            full = torch.arange(n_total * 60, dtype=torch.int16).reshape((n_total,) + shape)
        local = parallel.scatter_batch(full, n_total, shape, torch.int16, dev, src=root)
        a, b = parallel.shard_range(n_total, rank, world)
        expect = torch.arange(n_total * 60, dtype=torch.int16).reshape((n_total,) + shape)[a:b]
        ok_scatter = bool(torch.equal(local, expect))
        # every rank "filters" its shard (here: a deterministic function), root gathers
        res = (local.to(torch.int32) * 3 - 16).to(torch.int16)
        gathered = parallel.gather_batch(res, n_total, dst=root)
        ok_gather = True
        if rank == root:
            whole = torch.arange(n_total * 60, dtype=torch.int16).reshape((n_total,) + shape)
            ok_gather = bool(torch.equal(gathered, (whole.to(torch.int32) * 3 - 16).to(torch.int16)))
        else:
            ok_gather = gathered is None
        tmax = parallel.max_over_ranks(1.0 + rank, dev)
        tsum = parallel.sum_over_ranks(1.0 + rank, dev)
        ok_wire = all(d == torch.uint8 for d in posted) and (len(posted) > 0 or n_total < world)
        if rank == root:                   # one send and one receive per peer that owns at least one pair, nothing else
            peers = sum(1 for r in range(world) if r != root and parallel.shard_range(n_total, r, world)[1] > parallel.shard_range(n_total, r, world)[0])
            ok_wire = ok_wire and len(posted) == 2 * peers
        q.put((rank, ok_scatter and ok_wire, ok_gather, tmax, tsum))
    finally:
        dist.destroy_process_group()


def _run(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args[:-1] + (q,) + args[-1:]) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=240) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()                   # exactly the processes started above
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(r[0] for r in results) == list(range(world))
    return results


@pytest.mark.parametrize("world,n_total,root", [(2, 8, 0), (2, 5, 0), (2, 1, 0),
                                                (8, 8, 0), (8, 13, 0), (8, 64, 0), (8, 13, 3), (8, 5, 0)])
def test_scatter_gather(world, n_total, root):
    """Every pair lands at its own index exactly once, on the root of the exchange (root 3 once: the root need not be
    rank 0); shards are ragged for 13 pairs on 8 ranks and some ranks own nothing for 5 pairs on 8 ranks."""
    results = _run(_worker, world, n_total, root)
    for rank, ok_s, ok_g, tmax, tsum in results:
        assert ok_s and ok_g
        assert tmax == float(world) and tsum == world * (world + 1) / 2.0


def _pipe_worker(rank, world, port, n_total, n_sub, q, root=0):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        shapes = [(3, 5, 2), (3, 5), (3, 5)]
        dtypes = [torch.uint8, torch.int16, torch.int16]
        full = None
        if rank == root:
            g = torch.Generator().manual_seed(5)
            full = [torch.randint(0, 200, (n_total,) + s, generator=g).to(d) for s, d in zip(shapes, dtypes)]
        calls = []
        make_op, posted = dist.P2POp, []
        dist.P2POp = lambda op, tensor, *a, **k: (posted.append(tensor.dtype), make_op(op, tensor, *a, **k))[1]

        def process(v, a, b, o):            # stands in for DisparityWLSFilter.filter on one sub-batch
            calls.append(int(a.shape[0]))
            o.copy_((a.to(torch.int32) * 2 - b.to(torch.int32) + v[..., 0].to(torch.int32)).to(torch.int16))

        stats, out = parallel.pipelined_scatter_filter_gather(full, n_total, shapes, dtypes, (3, 5), torch.int16, dev,
                                                              process, n_sub, src=root)
        ok = all(d == torch.uint8 for d in posted)              # nothing 16-bit is posted (RCCL refuses it)
        if rank == root:
            exp = (full[1].to(torch.int32) * 2 - full[2].to(torch.int32) + full[0][..., 0].to(torch.int32)).to(torch.int16)
            ok = ok and bool(torch.equal(out, exp))
        else:
            ok = ok and out is None
        a, b = parallel.shard_range(n_total, rank, world)
        q.put((rank, ok, sum(calls) == b - a, stats["sub_batches"], stats["total_ms"] > 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,n_sub,root", [(2, 16, 4, 0), (2, 9, 4, 0), (2, 2, 4, 0), (2, 7, 1, 0),
                                                       (8, 64, 4, 0), (8, 13, 4, 0), (8, 8, 4, 0), (8, 64, 4, 5)])
def test_pipelined_scatter_filter_gather(world, n_total, n_sub, root):
    """sub-batch s+1 travels while sub-batch s is filtered; every pair is filtered exactly once and
    lands in the root's output at its own index, whatever the shard / sub-batch raggedness -- at world size 2 and at
    config 4's world size 8 (64 pairs = 8 per rank in 4 sub-batches of 2; 13 pairs: ragged shards, 1 sub-batch)."""
    results = _run(_pipe_worker, world, n_total, n_sub, root)
    subs = {r[3] for r in results}
    assert len(subs) == 1 and subs.pop() == max(1, min(n_sub, max(1, n_total // world)))
    for rank, ok, counted, _, timed in results:
        assert ok and counted and timed


def test_sub_ranges_tile_every_shard():
    for n in (1, 7, 64, 513):
        for world in (1, 2, 8):
            for n_sub in (1, 3, 4):
                for r in range(world):
                    a, b = parallel.shard_range(n, r, world)
                    spans = [parallel.sub_range(n, r, world, s, n_sub) for s in range(n_sub)]
                    assert spans[0][0] == a and spans[-1][1] == b
                    assert all(spans[i][1] == spans[i + 1][0] for i in range(n_sub - 1))
