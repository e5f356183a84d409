"""Where a plain filter() call from Python spends its host time (cProfile over 3000 calls of a config-5 frame):
python tools/profile_wrapper.py.  Round 4: wrapper 25 -> 10 us per call; the rest of a call is the C-ABI (seven launches)."""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
import addingdisparityfiltering_amd as adf
from addingdisparityfiltering_amd import synthetic
cfg = synthetic.CONFIGS[5]
dev = torch.device("cuda:0")
view, dl, dr = synthetic.make_artificial_batch_torch(4, cfg["W"], cfg["H"], 1, 1, cfg["rect_disparity"], dev)
out = torch.empty_like(dl)
f = adf.createDisparityWLSFilterGeneric(True); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(2)
roi = cfg["roi"]
for _ in range(20): f.filter(dl[0], view[0], out[0], dr[0], roi)
torch.cuda.synchronize()
N = 3000
t0 = time.perf_counter()
for i in range(N): f.filter(dl[0], view[0], out[0], dr[0], roi)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("issue us/call", (t1 - t0) / N * 1e6)
a, b, c, d = dl[0], view[0], out[0], dr[0]
t0 = time.perf_counter()
for i in range(N): f.filter(a, b, c, d, roi)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("issue us/call (no indexing)", (t1 - t0) / N * 1e6)
pr = cProfile.Profile(); pr.enable()
for i in range(N): f.filter(a, b, c, d, roi)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
