import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import addingdisparityfiltering_amd as adf
from addingdisparityfiltering_amd import synthetic
dev = torch.device("cuda:0")
cfg = synthetic.CONFIGS[3]
W, H, roi, radius = cfg["W"], cfg["H"], cfg["roi"], cfg["radius"]
view, dl, dr = synthetic.make_artificial_batch_torch(64, W, H, 3, synthetic.seed_for(3, 0), cfg["rect_disparity"], dev)
out = torch.empty((64, H, W), dtype=torch.int16, device=dev)
f = adf.createDisparityWLSFilterGeneric(True)
f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius); f.setSolver(adf.SOLVER_WAVE)
for prof in (True, False, True, False):
    f.enableProfiling(prof)
    for _ in range(3): f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f.filter(dl, view, out, dr, roi)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    if prof: f.readProfile()
    print("per-kernel events %s: %.3f ms per step" % ("on " if prof else "off", ms))
