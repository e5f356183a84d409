"""Times the device semi-global matcher (left view, then the right-view matcher of createRightMatcher) on synthetic
pairs: python tools/sgbm_time.py [W H ndisp block n channels mode]   (mode: 2 = MODE_SGBM_3WAY (default), 0 = MODE_SGBM, 1 = MODE_HH)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import addingdisparityfiltering_amd as adf

W, H, nd, bs, n, cn, mode = (int(v) for v in (sys.argv[1:8] + ["3840", "2160", "256", "3", "2", "1", "2"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
shape = (n, H, W + 64) + ((cn,) if cn > 1 else ())
base = rng.integers(0, 256, shape, dtype=np.uint8)
left = torch.from_numpy(np.ascontiguousarray(base[:, :, 32:32 + W])).cuda()
right = torch.from_numpy(np.ascontiguousarray(np.roll(base, -9, 2)[:, :, 32:32 + W])).cuda()
lm = adf.StereoSGBM.create(0, nd, bs)
lm.setP1(24 * bs * bs); lm.setP2(96 * bs * bs); lm.setPreFilterCap(63); lm.setMode(mode)
wls = adf.createDisparityWLSFilter(lm)                 # samples/disparity_filtering.cpp:166-172
rm = adf.createRightMatcher(lm)
dl = torch.empty((n, H, W), dtype=torch.int16, device="cuda")
dr = torch.empty_like(dl)
lm.compute(left, right, dl); rm.compute(right, left, dr)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
reps = 2
tl = tr = 0.0
for _ in range(reps):
    e[0].record(); lm.compute(left, right, dl); e[1].record(); rm.compute(right, left, dr); e[2].record()
    torch.cuda.synchronize()
    tl += e[0].elapsed_time(e[1]); tr += e[1].elapsed_time(e[2])
tl /= reps; tr /= reps
px = n * W * H
print("semi-global matcher (%s): %dx%dx%d ndisp %d block %d, %d pairs: left %.2f ms + right %.2f ms = %.3f ms/pair both views "
      "(%.2f Gpx/s, %.1f G(px*disp)/s per view)" % ({2: "3-way", 0: "5 paths", 1: "8 paths"}[mode], W, H, cn, nd, bs, n, tl, tr, (tl + tr) / n, px / (tl + tr) / 1e6,
                                                    2 * px * nd / (tl + tr) / 1e6))
