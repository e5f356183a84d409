"""The reference's own FGS perf test on the device (perf_fgs_filter.cpp:55-76): fastGlobalSmootherFilter(guide, src,
lambda, sigma) -- filter creation (edge weights) AND one filter call per cycle, a fresh lambda / sigma every cycle --
for guide 8UC1 / 8UC3 x src 8UC1 / 8UC3 / 16SC1 / 16SC3 / 32FC1 / 32FC3, device tensors in, device tensor out.
Also times filter() alone on a filter created once.   python tools/fgs_time.py [W H cycles]   (default 1280 720 10)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf

W, H, cycles = (int(v) for v in (sys.argv[1:4] + ["1280", "720", "10"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
dev = torch.device("cuda", 0)
dts = {"8U": torch.uint8, "16S": torch.int16, "32F": torch.float32}
print("fastGlobalSmootherFilter on %dx%d, %d cycles per case (perf_fgs_filter.cpp)" % (W, H, cycles))
for gch in (1, 3):
    guide = torch.from_numpy(rng.integers(0, 256, (H, W, gch) if gch > 1 else (H, W), dtype=np.uint8)).to(dev)
    for depth in ("8U", "16S", "32F"):
        for cn in (1, 3):
            shape = (H, W, cn) if cn > 1 else (H, W)
            if depth == "32F":
                src = torch.from_numpy(rng.random(shape, dtype=np.float32) * 255).to(dev)
            else:
                src = torch.from_numpy(rng.integers(0, 256, shape).astype(np.uint8 if depth == "8U" else np.int16)).to(dev)
            dst = torch.empty_like(src)
            lam = [float(rng.uniform(500.0, 10000.0)) for _ in range(cycles + 2)]
            sig = [float(rng.uniform(1.0, 100.0)) for _ in range(cycles + 2)]
            for k in range(2):                                   # warm-up (WARMUP_RNG)
                adf.fastGlobalSmootherFilter(guide, src, lam[k], sig[k], dst=dst)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(cycles):
                adf.fastGlobalSmootherFilter(guide, src, lam[2 + k], sig[2 + k], dst=dst)
            torch.cuda.synchronize()
            one_shot = (time.perf_counter() - t0) / cycles * 1e3
            f = adf.createFastGlobalSmootherFilter(guide, 8000.0, 1.5)
            f.filter(src, dst); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(cycles):
                f.filter(src, dst)
            torch.cuda.synchronize()
            reuse = (time.perf_counter() - t0) / cycles * 1e3
            print("guide 8UC%d src %sC%d: create + filter %.3f ms per call (%.2f Gpx/s), filter alone %.3f ms (%.2f Gpx/s)"
                  % (gch, depth, cn, one_shot, W * H / one_shot / 1e6, reuse, W * H / reuse / 1e6), flush=True)
            del f
