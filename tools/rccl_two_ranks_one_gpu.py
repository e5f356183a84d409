"""Can RCCL run two ranks on ONE GPU (a rehearsal of the N > 1 path on the one-GPU box)?  Spawns two processes on
cuda:0, backend nccl, tries an all-reduce and the scatter / gather of parallel.py.  Prints what happened; a refusal
("Duplicate GPU detected") is an answer too.  python tools/rccl_two_ranks_one_gpu.py
Answer on this pool (RCCL 2.26.6, round 3): refused -- "Duplicate GPU detected : rank 1 and rank 0 both on CUDA device
f1000": ranks cannot share a device, so RCCL runs here as a group of one only (tests/test_gpu_rccl_one_rank.py)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    import datetime
    import torch.distributed as dist
    from addingdisparityfiltering_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=60))
        t = torch.tensor([1.0 + rank], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        print("rank %d: all_reduce -> %s" % (rank, t.item()), flush=True)
        n = 6
        full = torch.arange(n * 1000, dtype=torch.int16, device=dev).reshape(n, 10, 100) if rank == 0 else None
        local = parallel.scatter_batch(full, n, (10, 100), torch.int16, dev)
        out = parallel.gather_batch((local + 1).contiguous(), n)
        torch.cuda.synchronize()
        if rank == 0:
            print("rank 0: scatter/gather equal:", bool(torch.equal(out, full + 1)), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:400]), flush=True)


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
