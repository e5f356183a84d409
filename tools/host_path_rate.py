"""PCIe-inclusive rate of the host-pointer entry point (adf_wls_filter_host through the Python mirror, pageable numpy
buffers): python tools/host_path_rate.py [pairs]   (BASELINE config 3 geometry)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as xi
from addingdisparityfiltering_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
view, dl, dr, roi, radius = synthetic.make_config_example(3)
views = np.ascontiguousarray(np.broadcast_to(view, (n,) + view.shape))
dls = np.ascontiguousarray(np.broadcast_to(dl, (n,) + dl.shape))
drs = np.ascontiguousarray(np.broadcast_to(dr, (n,) + dr.shape))
out = np.empty_like(dls)
wls = xi.createDisparityWLSFilterGeneric(True)
wls.setLambda(8000.0); wls.setSigmaColor(1.5); wls.setDepthDiscontinuityRadius(radius)
wls.filter(dls, views, out, drs, roi)
t = time.time(); reps = 3
for _ in range(reps):
    wls.filter(dls, views, out, drs, roi)
dt = (time.time() - t) / reps
H, W = dl.shape
print("host-pointer path, %d pairs of %dx%d per call (pageable numpy in and out): %.1f ms per call = %.2f ms/pair = %.2f Gpx/s; "
      "bytes over PCIe per pair: %.1f MB in, %.1f MB out" % (n, W, H, dt * 1e3, dt * 1e3 / n, n * W * H / dt / 1e9,
                                                        (view.nbytes + dl.nbytes + dr.nbytes) / 1e6, dl.nbytes / 1e6))
