"""The sample's ground-truth evaluation (samples/disparity_filtering.cpp:130-283) replayed on the stereo pair the
reference itself holds WITH a ground truth: modules/stereo/testdata/imL2l.bmp / imL2.bmp / groundtruth.bmp (Tsukuba,
384x288, disparity * 16; copies under tests/golden/, provenance in its README).  For every producer and filter mode of
the sample -- algo bm / sgbm, filter wls_conf (full-size and the default down-scaled views) / wls_no_conf -- it reports
what the sample prints: MSE and percent of bad pixels against the ground truth, before and after filtering, inside the
ROI (SAMPLE:268-283).  Nothing here pins bits (the reference publishes no numbers for this pair); what it pins is the
behaviour the tutorial promises (tutorials/disparity_filtering.markdown: the filtered map is closer to the truth than the
raw one), through every stage this repository builds: matcher (N4), down-scaled path (N1), confidence + filter
(A1-A12), evaluation utilities (N3).

    python tests/sample_evaluation.py [--hip]      prints the table (oracle pipeline; --hip: the device pipeline too)

Test infrastructure: tests/test_sample_evaluation.py gates on the bars below."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from tutorial_replay import half_size  # noqa: E402  (cv::resize 0.5 INTER_LINEAR on 8U = 2x2 mean, round half up)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MAX_DISP, LAMBDA, SIGMA = 16, 8000.0, 1.5              # test_block_matching.cpp:128 (16 disparities); SAMPLE:60-61 defaults
# (algo, window): the sample's defaults are bm 7 (SAMPLE:62 with the tutorial's command line) and sgbm 3 (SAMPLE:108-109)
CASES = [("bm", 7), ("bm", 9), ("sgbm", 3), ("sgbm", 5)]
MODES = ["wls_conf", "wls_conf_downscaled", "wls_no_conf"]


def load_fixture():
    from PIL import Image
    left = np.array(Image.open(os.path.join(GOLDEN, "stereo_left.bmp")).convert("L"))
    right = np.array(Image.open(os.path.join(GOLDEN, "stereo_right.bmp")).convert("L"))
    gt8 = np.array(Image.open(os.path.join(GOLDEN, "stereo_groundtruth.bmp")).convert("L"))
    # the file holds disparity * 16; 0 = unknown, which readGT / the metrics treat as "no ground truth" (DF.cpp:460,477)
    gt = np.where(gt8 == 0, 16320, gt8.astype(np.int32)).astype(np.int16)
    return left, right, gt


def plan(algo, wsize, mode, size):
    """What the sample sets up for (algo, filter): matcher sizes, ROI in the maps' coordinates, radius (SAMPLE:130-252)."""
    w, h = size
    nd = MAX_DISP
    if mode == "wls_conf_downscaled":                    # SAMPLE:135-141
        nd = MAX_DISP // 2
        if nd % 16:
            nd += 16 - nd % 16
        w, h = w // 2, h // 2
    w2 = wsize // 2
    if mode == "wls_no_conf":                            # computeROI, SAMPLE:333-350
        maxd = nd - 1
        roi = (maxd + w2, w2, (w - w2) - (maxd + w2), (h - w2) - w2)
    elif algo == "bm":                                   # createDisparityWLSFilter, DF.cpp:399-403
        roi = (nd + w2, w2, w - nd - 2 * w2, h - 2 * w2)
    else:                                                # DF.cpp:408-410
        roi = (nd, 0, w - nd, h)
    radius = int(math.ceil((0.33 if algo == "bm" else 0.5) * wsize))
    return nd, roi, radius


def evaluate_oracle(algo, wsize, mode, fixture, threads=8):
    import oracle
    left, right, gt = fixture
    H, W = left.shape
    nd, roi, radius = plan(algo, wsize, mode, (W, H))
    down = mode == "wls_conf_downscaled"
    lm, rm = (half_size(left), half_size(right)) if down else (left, right)

    def match(a, b, md):
        if algo == "bm":
            return oracle.bm_compute(a, b, nd, wsize, md)
        return oracle.sgbm_compute(a, b, nd, wsize, md, P1=24 * wsize * wsize, P2=96 * wsize * wsize, prefilter_cap=63)

    dl = match(lm, rm, 0)
    conf = mode != "wls_no_conf"
    p = oracle.default_params(lambda_=LAMBDA, sigma_color=SIGMA, disc_radius=radius, threads=threads, use_confidence=int(conf))
    if conf:
        dr = match(rm, lm, -nd + 1)                      # createRightMatcher, DF.cpp:421-446
        if down:
            out, _ = oracle.wls_filter_scaled(dl, left, dr, roi, p)
            raw = oracle.resize_linear(dl, (W, H), 2.0)  # SAMPLE:199-200: the raw map upscaled for the comparison
            roi = tuple(2 * v for v in roi)              # SAMPLE:201
        else:
            out, _ = oracle.wls_filter(dl, left, dr, roi, p)
            raw = dl
    else:
        out, _ = oracle.wls_filter(dl, left, None, roi, p, want_conf=False)
        raw = dl
    return metrics(oracle.compute_mse, oracle.bad_pixel_percent, gt, raw, out, roi), raw, out, roi


def evaluate_hip(algo, wsize, mode, fixture):
    import torch

    import addingdisparityfiltering_amd as xi
    left, right, gt = fixture
    H, W = left.shape
    nd, roi, radius = plan(algo, wsize, mode, (W, H))
    down = mode == "wls_conf_downscaled"
    lm, rm = (half_size(left), half_size(right)) if down else (left, right)
    dev = torch.device("cuda:0")
    tl, tr, tview = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (lm, rm, left))
    if algo == "bm":
        matcher = xi.StereoBM.create(nd, wsize)                                          # SAMPLE:151
    else:
        matcher = xi.StereoSGBM.create(0, nd, wsize)                                     # SAMPLE:166-172
        matcher.setP1(24 * wsize * wsize); matcher.setP2(96 * wsize * wsize)
        matcher.setPreFilterCap(63); matcher.setMode(xi.StereoSGBM.MODE_SGBM_3WAY)
    if mode != "wls_no_conf":
        wls = xi.createDisparityWLSFilter(matcher)                                       # SAMPLE:152
        right_matcher = xi.createRightMatcher(matcher)                                   # SAMPLE:153
        dl = matcher.compute(tl, tr)
        dr = right_matcher.compute(tr, tl)
        wls.setLambda(LAMBDA); wls.setSigmaColor(SIGMA)
        out = wls.filter(dl, tview, None, dr)                                            # SAMPLE:189
        got_roi = tuple(wls.getROI())                                                    # SAMPLE:194
        assert got_roi == roi and wls.getDepthDiscontinuityRadius() == radius
        raw = dl
        if down:
            raw = xi_resize_raw(dl.cpu().numpy(), (W, H))
            roi = tuple(2 * v for v in roi)
    else:
        if algo == "bm":
            matcher.setTextureThreshold(0); matcher.setUniquenessRatio(0)                # SAMPLE:216-217
        else:
            matcher.setUniquenessRatio(0); matcher.setDisp12MaxDiff(1000000); matcher.setSpeckleWindowSize(0)   # SAMPLE:232-234
        wls = xi.createDisparityWLSFilterGeneric(False)                                  # SAMPLE:221
        wls.setDepthDiscontinuityRadius(radius)
        dl = matcher.compute(tl, tr)
        wls.setLambda(LAMBDA); wls.setSigmaColor(SIGMA)
        out = wls.filter(dl, tview, None, None, roi)                                     # SAMPLE:253
        raw = dl
    torch.cuda.synchronize()
    raw = raw.cpu().numpy() if hasattr(raw, "cpu") else raw
    out = out.cpu().numpy()
    return metrics(xi.computeMSE, xi.computeBadPixelPercent, gt, raw, out, roi), raw, out, roi


def xi_resize_raw(dl, size):
    """SAMPLE:199-200 for the comparison only (the sample's own host-side upscale of the raw map): cv::resize of the int16
    map, then * 2 -- the oracle's statement of resize; not a product call."""
    import oracle
    return oracle.resize_linear(dl, size, 2.0)


def metrics(mse, bad, gt, raw, out, roi):
    return dict(mse_before=float(mse(gt, raw, roi)), mse_after=float(mse(gt, out, roi)),
                bad_before=float(bad(gt, raw, roi)), bad_after=float(bad(gt, out, roi)))


def fmt(algo, wsize, mode, m, roi):
    return "%-4s w=%d  %-20s ROI %-20s MSE %6.3f -> %6.3f   bad pixels %5.2f %% -> %5.2f %%" % (
        algo, wsize, mode, str(tuple(roi)), m["mse_before"], m["mse_after"], m["bad_before"], m["bad_after"])


def main():
    fx = load_fixture()
    hip = "--hip" in sys.argv
    print("# the sample's evaluation on the reference's Tsukuba fixture (384x288, 16 disparities, lambda 8000, sigma 1.5): oracle pipeline")
    for algo, w in CASES:
        for mode in MODES:
            m, _, _, roi = evaluate_oracle(algo, w, mode, fx)
            print(fmt(algo, w, mode, m, roi))
    if hip:
        print("# the same through the device pipeline (matcher, filter and metrics on the GPU)")
        for algo, w in CASES:
            for mode in MODES:
                m, _, _, roi = evaluate_hip(algo, w, mode, fx)
                print(fmt(algo, w, mode, m, roi))


if __name__ == "__main__":
    main()
