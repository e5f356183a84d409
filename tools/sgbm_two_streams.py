"""Measurement helper: the left-view and right-view semi-global matchers of one pair on TWO streams (the cost kernel of one
view is bound by vector issue, the path kernels of the other by memory) against one after the other.
    python tools/sgbm_two_streams.py [W H ndisp block n]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf

W, H, nd, bs, n = (int(v) for v in (sys.argv[1:6] + ["3840", "2160", "256", "3", "2"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (n, H, W + 64), dtype=np.uint8)
left = torch.from_numpy(np.ascontiguousarray(base[:, :, 32:32 + W])).cuda()
right = torch.from_numpy(np.ascontiguousarray(np.roll(base, -9, 2)[:, :, 32:32 + W])).cuda()
lm = adf.StereoSGBM.create(0, nd, bs)
lm.setP1(24 * bs * bs); lm.setP2(96 * bs * bs); lm.setPreFilterCap(63); lm.setMode(adf.StereoSGBM.MODE_SGBM_3WAY)
wls = adf.createDisparityWLSFilter(lm)
rm = adf.createRightMatcher(lm)
dl = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
dr = torch.empty_like(dl)
lm.compute(left, right, dl); rm.compute(right, left, dr)
torch.cuda.synchronize()
ref_l, ref_r = dl.clone(), dr.clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("one stream", "two streams", "one stream", "two streams"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        if mode == "one stream":
            lm.compute(left, right, dl); rm.compute(right, left, dr)
        else:
            with torch.cuda.stream(s1):
                lm.compute(left, right, dl)
            with torch.cuda.stream(s2):
                rm.compute(right, left, dr)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    print("%-12s %8.3f ms per %d pair(s) both views = %.3f ms/pair   (maps identical: %s)" % (mode, ms, n, ms / n, bool(torch.equal(dl, ref_l) and torch.equal(dr, ref_r))))
