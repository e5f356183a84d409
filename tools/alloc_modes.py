"""Is the 64 x 4K step's speed a property of WHERE the handle's workspace landed?  One process, the same inputs, the same
(NULL) stream; several handles alive at once (so their workspaces have different addresses), each timed alone, twice.
python tools/alloc_modes.py [handles]"""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hs = []
for k in range(K):
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
    f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize()
    hs.append(f)


def t(f, reps=8):
    f.filter(dl, view, out, dr, roi); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


free, total = torch.cuda.mem_get_info(dev)
print("%d handles alive (%.1f GiB in use)" % (K, (total - free) / 2**30))
for rnd in range(2):
    print("round %d: " % rnd + "  ".join("h%d %.3f" % (k, t(f)) for k, f in enumerate(hs)), flush=True)
