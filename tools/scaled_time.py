"""The down-scaled path (DF.cpp:224-227, 239-247, 268-277; the sample's default: matcher on half-size views, filter on
the full view): n pairs of W x H views with (W/2) x (H/2) disparity maps.  python tools/scaled_time.py [W H n scale]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf
from addingdisparityfiltering_amd import synthetic

W, H, n, sc = (int(v) for v in (sys.argv[1:5] + ["3840", "2160", "64", "2"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
dW, dH = W // sc, H // sc
nd = 256 // sc
view, _, _ = synthetic.make_artificial_batch_torch(n, W, H, 3, 1, 64, dev)
_, dl, dr = synthetic.make_artificial_batch_torch(n, dW, dH, 3, 1, 64 // sc, dev)
out = torch.empty((n, H, W), dtype=torch.int16, device=dev)
roi = (nd, 0, dW - nd, dH)
for radius in (2, 5):
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
    for _ in range(2):
        f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    f.enableProfiling(True)
    f.filter(dl, view, out, dr, roi); torch.cuda.synchronize()
    prof = f.readProfile()
    print("down-scaled path: %d pairs, views %dx%d, maps %dx%d, radius %d: %.3f ms per call = %.2f Gpx/s (view pixels); path flags %d"
          % (n, W, H, dW, dH, radius, ms, n * W * H / ms / 1e6, f.getLastPath()))
    print("   kernels (ms per call):", {k: round(v["total_ms"], 3) for k, v in prof.items()})
