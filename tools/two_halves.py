"""Scheduling experiment: the 64 x 4K step as ONE call against the same pairs dealt to K handles on K streams (each runs
the whole pipeline on its share; kernels of different kinds then overlap).  python tools/two_halves.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf
from addingdisparityfiltering_amd import synthetic

cfg = synthetic.CONFIGS[3]
dev = torch.device("cuda:0")
N = 64
view, dl, dr = synthetic.make_artificial_batch_torch(N, cfg["W"], cfg["H"], cfg["channels"], synthetic.seed_for(3, 0), cfg["rect_disparity"], dev)
out = torch.empty_like(dl)
roi, radius = cfg["roi"], cfg["radius"]


def make():
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
    return f


def run(K, reps=10):
    hs = [make() for _ in range(K)]
    ss = [torch.cuda.Stream(device=dev) for _ in range(K)]
    per = N // K

    def step():
        for k in range(K):
            a, b = k * per, (k + 1) * per
            with torch.cuda.stream(ss[k]):
                hs[k].filter(dl[a:b], view[a:b], out[a:b], dr[a:b], roi)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("K = %d handles / streams, %2d pairs each: %.3f ms per 64 pairs = %.1f Gpx/s" % (K, per, ms, N * cfg["W"] * cfg["H"] / ms / 1e6), flush=True)
    return out.clone()


ref = run(1)
for K in (2, 4, 1, 2):
    o = run(K)
    assert torch.equal(o, ref)
