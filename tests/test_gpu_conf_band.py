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


def _random_aligned_case(rng):
    """Geometry the one-sweep kernel takes: frame width, ROI x and ROI width multiples of 4, radius 1..4."""
    rw = 4 * int(rng.integers(2, 260)); rh = int(rng.integers(5, 140))
    rx = 4 * int(rng.integers(0, 40)); ry = int(rng.integers(0, 9))
    W = rx + rw + 4 * int(rng.integers(0, 30)); H = ry + rh + int(rng.integers(0, 9))
    radius = int(rng.integers(1, 5))
    if rh <= radius:
        rh = radius + 1; H = max(H, ry + rh)
    kind = str(rng.choice(["scene", "wide", "wild"]))
    return W, H, (rx, ry, rw, rh), radius, kind, int(rng.integers(0, 80))


@pytest.mark.parametrize("seed", range(40))
def test_random_aligned_geometries(adf, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    W, H, roi, radius, kind, thresh = _random_aligned_case(rng)
    _check(adf, oracle, W, H, roi, radius, kind, 7000 + seed, thresh)
