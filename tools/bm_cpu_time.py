"""CPU time of the block-matcher oracle (single thread) beside tools/bm_time.py: python tools/bm_cpu_time.py [W H ndisp wsz]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import oracle

W, H, nd, wsz = (int(v) for v in (sys.argv[1:5] + ["1920", "1080", "160", "15"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (H, W + 64), dtype=np.uint8)
left = np.ascontiguousarray(base[:, 32:32 + W]); right = np.ascontiguousarray(np.roll(base, -9, 1)[:, 32:32 + W])
t = time.time()
oracle.bm_compute(left, right, nd, wsz, 0)
oracle.bm_compute(right, left, nd, wsz, -nd + 1)
dt = time.time() - t
print("oracle matcher both views (1 thread): %dx%d ndisp %d block %d: %.2f s  (%.2f Mpx/s)" % (W, H, nd, wsz, dt, W * H / dt / 1e6))
