"""Do some device allocations stream slower than others?  Eight 8-GiB blocks from hipMalloc (through torch), each timed
alone: fill (write-only), read-only reduction, in-place read-modify-write.  python tools/alloc_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
blocks = [torch.empty(8 * 2**30 // 4, dtype=torch.float32, device=dev) for _ in range(8)]
for b in blocks:
    b.zero_()
torch.cuda.synchronize()


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(2):
    print("round %d" % rnd)
    print("  fill  (ms): " + "  ".join("%.3f" % t(lambda b=b: b.fill_(1.0)) for b in blocks))
    print("  rmw   (ms): " + "  ".join("%.3f" % t(lambda b=b: b.add_(1.0)) for b in blocks))
    print("  read  (ms): " + "  ".join("%.3f" % t(lambda b=b: b.view(torch.int32).bitwise_and(1).any()) for b in blocks[:2]), flush=True)
print("addresses mod 1 GiB (MiB): " + "  ".join("%d" % ((b.data_ptr() % 2**30) // 2**20) for b in blocks))
