import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import addingdisparityfiltering_amd as adf
import oracle
from addingdisparityfiltering_amd import synthetic
view, dl, dr, _ = synthetic.make_artificial_example(96, 80, 3, seed=21)
for roi in ((13, 7, 70, 60), (0, 0, 96, 80), (12, 7, 70, 60), (13, 0, 70, 60), (13, 7, 64, 60)):
    p = oracle.default_params(threads=4, sigma_color=2.0, disc_radius=3)
    exp, exp_conf = oracle.wls_filter(dl, view, dr, roi, p)
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setSolver(adf.SOLVER_WAVE); f.setSigmaColor(2.0); f.setDepthDiscontinuityRadius(3)
    got = f.filter(dl, view, None, dr, roi)
    d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    ys, xs = np.nonzero(d > 1)
    print(roi, "merge", os.environ.get("ADF_MERGE_SMALL"), "path", f.getLastPath() if hasattr(f, "getLastPath") else None, "max", d.max(), "n>1", len(ys),
          "rows", (ys.min(), ys.max()) if len(ys) else None, "cols", (xs.min(), xs.max()) if len(xs) else None, "conf ok", np.array_equal(f.getConfidenceMap(), exp_conf))
