"""Placement probe (profiles/r04_allocation_placement.txt): a dummy block of DUMMY_GB GiB is allocated before the inputs or
between the inputs and the handle's workspace (DUMMY_WHERE=before_inputs | after_inputs), so that the workspace lands
elsewhere in device memory; the 64 x 4K step is then timed.  One process per setting:
    DUMMY_GB=90 DUMMY_WHERE=after_inputs python tools/placement_dummy.py"""
import os, sys, time
import torch
sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf
from addingdisparityfiltering_amd import synthetic
cfg = synthetic.CONFIGS[3]
dev = torch.device("cuda:0")
gb = float(os.environ.get("DUMMY_GB", "0"))
where = os.environ.get("DUMMY_WHERE", "before_inputs")
dummy = None
if gb > 0 and where == "before_inputs":
    dummy = torch.empty(int(gb * 2**30), dtype=torch.uint8, device=dev)
N = 64
view, dl, dr = synthetic.make_artificial_batch_torch(N, cfg["W"], cfg["H"], cfg["channels"], synthetic.seed_for(3, 0), cfg["rect_disparity"], dev)
out = torch.empty_like(dl)
if gb > 0 and where == "after_inputs":
    dummy = torch.empty(int(gb * 2**30), dtype=torch.uint8, device=dev)
f = adf.createDisparityWLSFilterGeneric(True)
f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(cfg["radius"])
for _ in range(3):
    f.filter(dl, view, out, dr, cfg["roi"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    f.filter(dl, view, out, dr, cfg["roi"])
torch.cuda.synchronize()
print("dummy %5.1f GB %-14s: %.3f ms" % (gb, where, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
