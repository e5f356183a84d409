"""Times the device block matcher (both views) on synthetic pairs: python tools/bm_time.py [W H ndisp wsz n [uniq texture]]
(uniq / texture: uniquenessRatio and textureThreshold of the LEFT matcher, default 0 as the filter factory sets them)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf

W, H, nd, wsz, n = (int(v) for v in (sys.argv[1:6] + ["3840", "2160", "256", "15", "4"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (n, H, W + 64), dtype=np.uint8)
left = torch.from_numpy(np.ascontiguousarray(base[:, :, 32:32 + W])).cuda()
right = torch.from_numpy(np.ascontiguousarray(np.roll(base, -9, 2)[:, :, 32:32 + W])).cuda()
lm = adf.StereoBM.create(nd, wsz)
lm.setTextureThreshold(0); lm.setUniquenessRatio(0)
rm = adf.createRightMatcher(lm)
if len(sys.argv) > 6:
    lm.setUniquenessRatio(int(sys.argv[6])); rm.setUniquenessRatio(int(sys.argv[6]))
if len(sys.argv) > 7:
    lm.setTextureThreshold(int(sys.argv[7])); rm.setTextureThreshold(int(sys.argv[7]))
dl = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
dr = torch.empty_like(dl)
for _ in range(2):
    lm.compute(left, right, dl); rm.compute(right, left, dr)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 3
e0.record()
for _ in range(reps):
    lm.compute(left, right, dl); rm.compute(right, left, dr)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
px = n * W * H
both = None
if len(sys.argv) <= 6:                       # the filter's configuration: also time both views in one launch
    for _ in range(2):
        lm.computeBoth(left, right, dl, dr)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lm.computeBoth(left, right, dl, dr)
    e1.record(); torch.cuda.synchronize()
    both = e0.elapsed_time(e1) / reps
print("matcher both views: %dx%d ndisp %d block %d, %d pairs: %.2f ms  (%.3f ms/pair, %.2f Gpx/s, %.1f G(px*disp)/s per view)" %
      (W, H, nd, wsz, n, ms, ms / n, px / ms / 1e6, 2 * px * nd / ms / 1e6) +
      ("" if both is None else "; one launch for both views: %.2f ms (%.3f ms/pair)" % (both, both / n)))
