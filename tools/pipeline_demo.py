"""The sample's flow (samples/disparity_filtering.cpp:151-189, 253-283) on the reference's own stereo fixture, every
stage on the device: left / right block matcher -> DisparityWLSFilter -> computeMSE / computeBadPixelPercent against
the fixture's ground truth, before and after filtering.  python tools/pipeline_demo.py [block_size] [bm|sgbm]
(sgbm: the sample's default producer, StereoSGBM in MODE_SGBM_3WAY with P1 = 24*w*w, P2 = 96*w*w, SAMPLE:166-172)"""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import addingdisparityfiltering_amd as xi

G = os.path.join(ROOT, "tests", "golden")
left = np.array(Image.open(os.path.join(G, "stereo_left.bmp")).convert("L"))
right = np.array(Image.open(os.path.join(G, "stereo_right.bmp")).convert("L"))
gt8 = np.array(Image.open(os.path.join(G, "stereo_groundtruth.bmp")).convert("L"))
GT = np.where(gt8 == 0, 16320, gt8.astype(np.int32)).astype(np.int16)       # the file holds disparity*16; 0 = unknown (DF.cpp:460)
wsize = int(sys.argv[1]) if len(sys.argv) > 1 else 9
algo = sys.argv[2] if len(sys.argv) > 2 else "bm"
max_disp = 16

tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
if algo == "sgbm":
    left_matcher = xi.StereoSGBM.create(0, max_disp, wsize)                 # SAMPLE:166
    left_matcher.setP1(24 * wsize * wsize); left_matcher.setP2(96 * wsize * wsize)
    left_matcher.setPreFilterCap(63); left_matcher.setMode(xi.StereoSGBM.MODE_SGBM_3WAY)
else:
    left_matcher = xi.StereoBM.create(max_disp, wsize)                      # SAMPLE:151
wls = xi.createDisparityWLSFilter(left_matcher)                             # SAMPLE:152
right_matcher = xi.createRightMatcher(left_matcher)                         # SAMPLE:153
left_disp = left_matcher.compute(tl, tr)                                    # SAMPLE:160
right_disp = right_matcher.compute(tr, tl)                                  # SAMPLE:161
wls.setLambda(8000.0); wls.setSigmaColor(1.5)                               # SAMPLE:186-187
filtered = wls.filter(left_disp, tl, None, right_disp)                      # SAMPLE:189
torch.cuda.synchronize()
ROI = wls.getROI()                                                          # SAMPLE:194
conf = wls.getConfidenceMap()
conf = conf.cpu().numpy() if hasattr(conf, "cpu") else conf
for name, d in (("raw  ", left_disp.cpu().numpy()), ("wls  ", filtered.cpu().numpy())):
    print("%s MSE %.3f   bad pixels (>= 1.5 px) %.2f %%" % (name, xi.computeMSE(GT, d, ROI), xi.computeBadPixelPercent(GT, d, ROI)))   # SAMPLE:268-283
print(algo, "block", wsize, "ROI", ROI, " mean confidence %.1f" % conf[ROI[1]:ROI[1] + ROI[3], ROI[0]:ROI[0] + ROI[2]].mean(),
      " solver", "wave" if wls.getLastSolver() == xi.SOLVER_WAVE else "exact")
