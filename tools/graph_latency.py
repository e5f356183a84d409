"""Single-pair latency of the filter (and of matcher + filter) with and without a captured HIP graph:
python tools/graph_latency.py [config]   (BASELINE config 2 = one 1920x1080 pair)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as xi
from addingdisparityfiltering_amd import synthetic

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
view, dl, dr, roi, radius = synthetic.make_config_example(cfg)
dev = torch.device("cuda:0")
tv, tl, tr = (torch.from_numpy(a).to(dev) for a in (view, dl, dr))
out = torch.empty_like(tl)
wls = xi.createDisparityWLSFilterGeneric(True)
wls.setLambda(8000.0); wls.setSigmaColor(1.5); wls.setDepthDiscontinuityRadius(radius)


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


plain = lambda: wls.filter(tl, tv, out, tr, roi)
plain(); torch.cuda.synchronize()
ref = out.clone()
t_plain = timed(plain)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plain()                                   # workspace exists before the capture
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    wls.filter(tl, tv, out, tr, roi)
out.zero_()
g.replay(); torch.cuda.synchronize()
same = bool((out == ref).all())
t_graph = timed(g.replay)
H, W = tl.shape
print("config %d (%dx%d, one pair): filter %.3f ms per call, %.3f ms per graph replay (identical output: %s)" % (cfg, W, H, t_plain, t_graph, same))
