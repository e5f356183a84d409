"""The one-sweep confidence stage (conf_band_kernel, csrc/conf_kernels.hip): both views' depth-discontinuity maps,
the left-right check and x255 with the right view's map held in LDS only (DF.cpp:197-210).  Integer / separately
rounded float work: the confidence map must equal the oracle's BIT FOR BIT for every geometry the kernel takes --
wave boundaries (248 output columns per wave), image-edge reflection inside one lane, bands, every radius -- and
for arbitrary disparities (the gather may land anywhere in the row)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _maps(rng, H, W, kind):
    if kind == "wild":                                    # any int16: gathers all over the row, most of them out of range
        dl = rng.integers(-32768, 32768, (H, W)).astype(np.int16)
        dr = rng.integers(-32768, 32768, (H, W)).astype(np.int16)
    elif kind == "wide":                                  # disparities up to the full width, consistent left/right
        d = rng.integers(0, 16 * W, (H, W))
        dl = d.astype(np.int16); dr = (-d + rng.integers(-40, 40, (H, W))).astype(np.int16)
    else:                                                 # plausible scene: smooth + steps + noise
        base = (rng.integers(0, 60, (H, 1)) * 16 + 16 * 20 * (np.arange(W)[None, :] > W // 2)).astype(np.int64)
        dl = np.clip(base + rng.normal(0, 6, (H, W)), -32768, 32767).astype(np.int16)
        dr = np.clip(-base + rng.normal(0, 6, (H, W)), -32768, 32767).astype(np.int16)
    return dl, dr


def _check(adf, oracle, W, H, roi, radius, kind, seed, thresh=24):
    rng = np.random.default_rng(seed)
    dl, dr = _maps(rng, H, W, kind)
    view = rng.integers(0, 255, (H, W), dtype=np.uint8)
    exp = oracle.confidence(dl, dr, roi, radius, thresh, threads=8)
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setDepthDiscontinuityRadius(radius); f.setLRCthresh(thresh)
    f.filter(dl, view, None, dr, roi)
    assert f.getLastSolver() == adf.SOLVER_WAVE
    # the kernels under test really ran: the one-sweep confidence kernel and the first row pass that reads its map
    # (ROIs with no more rows than the radius, or narrower than 8 columns, take the column-walking kernels instead)
    band = roi[3] > radius and roi[2] >= 8 and roi[2] > radius
    # (small calls take weights + confidence + fill as ONE launch: the same device code, PATH_MERGED_PREP on top)
    assert f.getLastPath() & ~adf.PATH_MERGED_PREP == (adf.PATH_CONF_BAND if band else 0) | adf.PATH_FUSED_FIRST_PASS, f.getLastPath()
    got = f.getConfidenceMap()
    assert np.array_equal(got, exp), (W, H, roi, radius, kind, int((got != exp).sum()))
    return f


# ROI widths around the wave / lane boundaries of the kernel (4 columns per lane, 248 output columns per wave)
@pytest.mark.parametrize("rw", [8, 12, 244, 248, 252, 256, 496, 500, 744, 1000, 1984, 2232, 3584, 3964, 3968])
@pytest.mark.parametrize("radius", [1, 2, 3, 4])
def test_widths_and_radii(adf, oracle, rw, radius):
    H = 37
    W = rw + 8
    _check(adf, oracle, W, H, (4, 0, rw, H), radius, "scene", rw * 10 + radius)


# Radius 5..8 (round 3): two halo lanes per side, 240 output columns per wave.  5 is createDisparityWLSFilterGeneric's
# default and the StereoBM factory's value for its default block sizes (DF.cpp:155, 402); widths around the wave
# boundaries, not multiples of 4, up to the widest row 16 waves cover.
@pytest.mark.parametrize("rw", [9, 13, 236, 239, 240, 241, 244, 480, 483, 721, 1000, 2399, 2401, 3570, 3838, 3840])
@pytest.mark.parametrize("radius", [5, 6, 7, 8])
def test_widths_and_large_radii(adf, oracle, rw, radius):
    H = 41
    W = rw + 11
    _check(adf, oracle, W, H, (7, 1, rw, H - 2), radius, "scene", rw * 10 + radius)


@pytest.mark.parametrize("kind", ["wild", "wide"])
@pytest.mark.parametrize("radius", [5, 8])
def test_arbitrary_disparities_large_radii(adf, oracle, kind, radius):
    _check(adf, oracle, 523, 70, (9, 3, 501, 60), radius, kind, 500 + radius)


def test_bm_factory_geometry(adf, oracle):
    """The ROI and radius createDisparityWLSFilter derives from the sample's full-size StereoBM (block 15, DF.cpp:401-402:
    x = numDisparities + 7, radius ceil(0.33 * 15) = 5) on a 4K-wide strip: odd ROI x, width 3570 = 4*892 + 2."""
    _check(adf, oracle, 3840, 64, (263, 7, 3570, 50), 5, "scene", 4242)


@pytest.mark.parametrize("kind", ["wild", "wide", "scene"])
@pytest.mark.parametrize("radius", [1, 2, 4])
def test_arbitrary_disparities(adf, oracle, kind, radius):
    _check(adf, oracle, 520, 70, (8, 3, 500, 60), radius, kind, 5 + radius)
    _check(adf, oracle, 1024, 33, (0, 0, 1024, 33), radius, kind, 50 + radius)


@pytest.mark.parametrize("rh", [2, 3, 4, 5, 9, 31, 32, 33, 64, 65, 100])
def test_heights_and_bands(adf, oracle, rh):
    """Fewer rows than the window, band boundaries (32-row bands for a single small image), rows above / below the ROI."""
    for radius in (2, 4):
        _check(adf, oracle, 300, rh + 5, (12, 2, 280, rh), radius, "scene", rh * 3 + radius)


def test_thresholds_and_batch(adf, oracle):
    import torch
    rng = np.random.default_rng(77)
    n, H, W, roi = 5, 90, 400, (16, 0, 384, 90)
    dls, drs = zip(*[_maps(rng, H, W, "scene") for _ in range(n)])
    dl, dr = np.stack(dls), np.stack(drs)
    view = rng.integers(0, 255, (n, H, W, 3), dtype=np.uint8)
    for thresh in (0, 1, 24, 1000):
        f = adf.createDisparityWLSFilterGeneric(True)
        f.setDepthDiscontinuityRadius(2); f.setLRCthresh(thresh)
        f.filter(torch.from_numpy(dl).cuda(), torch.from_numpy(view).cuda(), None, torch.from_numpy(dr).cuda(), roi)
        got = f.getConfidenceMap().cpu().numpy()
        for k in range(n):
            assert np.array_equal(got[k], oracle.confidence(dl[k], dr[k], roi, 2, thresh, threads=8)), (thresh, k)


def test_band_kernel_equals_two_kernel_stage(adf, oracle):
    """A/B: ADF_CONF_BAND=0 selects the previous stage (right map written to HBM, left kernel gathers it)."""
    rng = np.random.default_rng(3)
    H, W, roi = 120, 1300, (40, 4, 1248, 110)
    dl, dr = _maps(rng, H, W, "scene")
    view = rng.integers(0, 255, (H, W, 3), dtype=np.uint8)
    outs = []
    for flag in ("1", "0"):
        os.environ["ADF_CONF_BAND"] = flag
        try:
            f = adf.createDisparityWLSFilterGeneric(True)
        finally:
            del os.environ["ADF_CONF_BAND"]
        f.setDepthDiscontinuityRadius(3)
        out = f.filter(dl, view, None, dr, roi)
        outs.append((out, f.getConfidenceMap()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][1], oracle.confidence(dl, dr, roi, 3, 24, threads=8))


def _random_aligned_case(rng, unaligned=False):
    """Geometry of round 2's fast path: frame width, ROI x and ROI width multiples of 4, radius 1..4 -- or (round 3)
    anything: odd ROI x / width / frame width, radius 1..8."""
    q = 1 if unaligned else 4
    rw = q * int(rng.integers(2, 260) * (4 // q)) + (int(rng.integers(0, 4)) if unaligned else 0); rh = int(rng.integers(5, 140))
    rx = q * int(rng.integers(0, 40) * (4 // q)) + (int(rng.integers(0, 4)) if unaligned else 0); ry = int(rng.integers(0, 9))
    W = rx + rw + q * int(rng.integers(0, 30)); H = ry + rh + int(rng.integers(0, 9))
    radius = int(rng.integers(1, 9 if unaligned else 5))
    if unaligned:
        rw = max(rw, 9)
    W = max(W, rx + rw)
    if rh <= radius:
        rh = radius + 1; H = max(H, ry + rh)
    kind = str(rng.choice(["scene", "wide", "wild"]))
    return W, H, (rx, ry, rw, rh), radius, kind, int(rng.integers(0, 80))


@pytest.mark.parametrize("seed", range(40))
def test_random_aligned_geometries(adf, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    W, H, roi, radius, kind, thresh = _random_aligned_case(rng)
    _check(adf, oracle, W, H, roi, radius, kind, 7000 + seed, thresh)


@pytest.mark.parametrize("seed", range(40))
def test_random_unaligned_geometries(adf, oracle, seed):
    """Round 3: any ROI x / width / frame width and radius 1..8 take the one-sweep kernel and the fused first pass."""
    rng = np.random.default_rng(9000 + seed)
    W, H, roi, radius, kind, thresh = _random_aligned_case(rng, unaligned=True)
    _check(adf, oracle, W, H, roi, radius, kind, 9000 + seed, thresh)


@pytest.mark.parametrize("rx,rw,W,off", [(263, 3570, 3840, 0), (1, 9, 11, 1), (3, 250, 256, 3), (5, 1003, 1011, 1), (2, 4, 8, 0), (7, 5, 13, 1)])
def test_filtered_map_on_unaligned_rois(adf, oracle, rx, rw, W, off):
    """The whole call on ROIs the first row pass could not fuse before round 3 (ROI x, width not multiples of 4; disparity
    rows that start at an odd element of a wider device tensor, so the 8-byte loads of dL sit at 2-byte aligned
    addresses): confidence bit-exact, filtered map within the reference's own bar of the oracle, exact solver
    bit-exact, and the partial last vector of a row never leaks into the padding (same result as on a dense copy)."""
    import torch
    rng = np.random.default_rng(rx * 7 + rw)
    H = 45; roi = (rx, 2, rw, H - 5); radius = 5 if rw > 8 else 2
    dl, dr = _maps(rng, H, W, "scene")
    view = rng.integers(0, 255, (H, W, 3), dtype=np.uint8)
    p = oracle.default_params(threads=8, use_confidence=1, disc_radius=radius, sigma_color=1.5)
    p.lambda_ = 8000.0
    exp, exp_conf = oracle.wls_filter(dl, view, dr, roi, p)
    wide_l = torch.zeros((H, W + 6), dtype=torch.int16, device="cuda"); wide_r = torch.zeros_like(wide_l)
    wide_l[:, off:off + W] = torch.from_numpy(dl).cuda(); wide_r[:, off:off + W] = torch.from_numpy(dr).cuda()
    tv = torch.from_numpy(view).cuda()
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
    for solver in (adf.SOLVER_WAVE, adf.SOLVER_EXACT):
        f.setSolver(solver)
        out = f.filter(wide_l[:, off:off + W], tv, None, wide_r[:, off:off + W], roi)
        torch.cuda.synchronize()
        assert np.array_equal(f.getConfidenceMap().cpu().numpy(), exp_conf)
        d = np.abs(out.cpu().numpy().astype(np.int64) - exp.astype(np.int64))
        if solver == adf.SOLVER_EXACT:
            assert d.max() == 0
        else:
            assert f.getLastPath() & ~adf.PATH_MERGED_PREP == (adf.PATH_CONF_BAND if rw >= 8 else 0) | adf.PATH_FUSED_FIRST_PASS
            assert d.max() <= 1 and d.mean() <= 1 / 256.0, (d.max(), d.mean())
            dense = f.filter(torch.from_numpy(dl).cuda(), tv, None, torch.from_numpy(dr).cuda(), roi)
            assert torch.equal(dense, out)


@pytest.mark.parametrize("W,H,roi,radius,ch", [(1242, 375, (128, 0, 1114, 375), 2, 1), (1920, 1080, (160, 0, 1760, 1080), 2, 3),
                                               (640, 480, (71, 7, 562, 466), 5, 3), (300, 40, (3, 1, 290, 37), 4, 1), (100, 9, (0, 0, 100, 9), 1, 3)])
def test_merged_preparation_launch_equals_the_three_kernels(adf, oracle, W, H, roi, radius, ch):
    """One small frame per call: edge weights, confidence map and the fill outside the ROI are ONE launch (round 3,
    single-call latency).  Same device code as the three kernels: confidence, filtered map and the -16 fill must be
    identical to the ADF_MERGE_SMALL=0 handle's, and equal to the oracle's."""
    import torch
    rng = np.random.default_rng(W + H + radius)
    dl, dr = _maps(rng, H, W, "scene")
    view = rng.integers(0, 255, (H, W, ch) if ch > 1 else (H, W), dtype=np.uint8)
    tl, tr, tv = (torch.from_numpy(a).cuda() for a in (dl, dr, view))
    res = []
    for flag in ("1", "0"):
        os.environ["ADF_MERGE_SMALL"] = flag
        try:
            f = adf.createDisparityWLSFilterGeneric(True)
        finally:
            del os.environ["ADF_MERGE_SMALL"]
        f.setLambda(8000.0); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
        out = torch.full((H, W), 1234, dtype=torch.int16, device="cuda")
        f.filter(tl, tv, out, tr, roi)
        torch.cuda.synchronize()
        assert bool(f.getLastPath() & adf.PATH_MERGED_PREP) == (flag == "1")
        res.append((out.cpu().numpy(), f.getConfidenceMap().cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    p = oracle.default_params(threads=8, use_confidence=1, disc_radius=radius, sigma_color=1.5)
    p.lambda_ = 8000.0
    exp, exp_conf = oracle.wls_filter(dl, view, dr, roi, p)
    assert np.array_equal(res[0][1], exp_conf)
    d = np.abs(res[0][0].astype(np.int64) - exp.astype(np.int64))
    assert d.max() <= 1 and d.mean() <= 1 / 256.0


def test_batches_keep_the_two_stream_preparation(adf):
    """Above 2.5 Mpixels of ROI per call the three kernels stay separate launches (two streams overlap them)."""
    import torch
    rng = np.random.default_rng(5)
    n, H, W, roi = 8, 375, 1242, (128, 0, 1114, 375)
    dl = torch.from_numpy(rng.integers(-100, 2000, (n, H, W)).astype(np.int16)).cuda()
    view = torch.from_numpy(rng.integers(0, 255, (n, H, W), dtype=np.uint8)).cuda()
    f = adf.createDisparityWLSFilterGeneric(True)
    f.setDepthDiscontinuityRadius(2)
    f.filter(dl, view, None, -dl, roi)
    assert f.getLastPath() == adf.PATH_CONF_BAND | adf.PATH_FUSED_FIRST_PASS


@pytest.mark.gpu
def test_tall_single_band_is_cut_at_the_row_cap():
    """The band kernel's low-half sum only grows (conf_kernels.hip: ColSum::slide), so a band is at most 2048 output rows.
    ADF_CONF_BANDS_TOTAL=1 asks for ONE band per image (a schedule only hundreds of pairs per call would produce): a
    4300-row frame at the widest radius, disparities with large squares, must come out in several bands, bit for bit.
    A child process, because the switch is read once per process."""
    import os
    import subprocess
    import sys

    code = r'''
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import addingdisparityfiltering_amd as adf, oracle
rng = np.random.default_rng(2)
w, h, radius = 192, 4300, 8
dl = rng.integers(-32768, 32767, (h, w)).astype(np.int16)
dr = rng.integers(-32768, 32767, (h, w)).astype(np.int16)
dl[:, ::3] = 255 * 128 + 127           # squares whose low halves are close to the maximum
view = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
roi = (16, 0, w - 16, h)
p = oracle.default_params(threads=8, sigma_color=1.5, disc_radius=radius)
exp, exp_conf = oracle.wls_filter(dl, view, dr, roi, p)
f = adf.createDisparityWLSFilterGeneric(True)
f.setSolver(adf.SOLVER_WAVE); f.setSigmaColor(1.5); f.setDepthDiscontinuityRadius(radius)
got = f.filter(dl, view, None, dr, roi)
assert f.getLastPath() & adf.PATH_CONF_BAND
assert np.array_equal(f.getConfidenceMap(), exp_conf)
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADF_CONF_BANDS_TOTAL="1", ADF_MERGE_SMALL="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
