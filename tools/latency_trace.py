"""One pair per call, back to back, no per-kernel events (for a rocprofv3 kernel trace of the call's timeline):
python tools/latency_trace.py config [calls]"""
import sys

import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as xi
from addingdisparityfiltering_amd import synthetic

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
view, dl, dr, roi, radius = synthetic.make_config_example(cfg)
dev = torch.device("cuda:0")
tv, tl, tr = (torch.from_numpy(a).to(dev) for a in (view, dl, dr))
out = torch.empty_like(tl)
wls = xi.createDisparityWLSFilterGeneric(True)
wls.setLambda(8000.0); wls.setSigmaColor(1.5); wls.setDepthDiscontinuityRadius(radius)
for _ in range(20):
    wls.filter(tl, tv, out, tr, roi)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(calls):
    wls.filter(tl, tv, out, tr, roi)
e1.record(); torch.cuda.synchronize()
print("config %d: %d calls back to back, %.2f us per call (host-issued, no graph)" % (cfg, calls, e0.elapsed_time(e1) * 1e3 / calls))
