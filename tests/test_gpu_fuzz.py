"""Randomised parity sweep on the GPU: shapes, ROIs, channels, radii and filter parameters drawn at
random (fixed seed), every case checked against the CPU oracle -- bit-exact for the exact solver and the
confidence map, within the reference's reproducibility bar for the wave solver."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng):
    w = int(rng.integers(8, 700)); h = int(rng.integers(8, 500))
    ch = int(rng.choice([1, 3]))
    rx = int(rng.integers(0, max(1, w // 3))); ry = int(rng.integers(0, max(1, h // 4)))
    rw = int(rng.integers(2, w - rx + 1)); rh = int(rng.integers(2, h - ry + 1))
    view = rng.integers(0, 256, (h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)
    if rng.random() < 0.5:                                    # piecewise-smooth guide: strong coupling
        view = (view // 64 * 64).astype(np.uint8)
    base = rng.integers(0, 16 * 40)
    dl = (base + rng.normal(0, 30, (h, w))).clip(-32768, 32767).astype(np.int16)
    dr = (-base + rng.normal(0, 30, (h, w))).clip(-32768, 32767).astype(np.int16)
    return dict(w=w, h=h, ch=ch, roi=(rx, ry, rw, rh), view=view, dl=dl, dr=dr,
                lam=float(rng.uniform(0, 20000)), sigma=float(rng.uniform(0.3, 60.0)),
                radius=int(rng.integers(0, 12)), thresh=int(rng.integers(1, 64)),
                num_iter=int(rng.integers(1, 5)), atten=float(rng.choice([0.25, 0.5, 1.0])),
                use_conf=bool(rng.random() < 0.7))


@pytest.mark.parametrize("seed", range(24))
def test_random_case(adf, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    c = _case(rng)
    p = oracle.default_params(threads=8, use_confidence=int(c["use_conf"]), sigma_color=c["sigma"],
                              disc_radius=c["radius"], lrc_thresh=c["thresh"], num_iter=c["num_iter"],
                              lambda_attenuation=c["atten"])
    p.lambda_ = c["lam"]
    exp, exp_conf = oracle.wls_filter(c["dl"], c["view"], c["dr"] if c["use_conf"] else None, c["roi"], p)
    f = adf.createDisparityWLSFilterGeneric(c["use_conf"])
    f.setSolver(adf.SOLVER_EXACT)
    f.setLambda(c["lam"]); f.setSigmaColor(c["sigma"]); f.setDepthDiscontinuityRadius(c["radius"])
    f.setLRCthresh(c["thresh"]); f.setFGSParams(c["atten"], c["num_iter"])
    got = f.filter(c["dl"], c["view"], None, c["dr"] if c["use_conf"] else None, c["roi"])
    assert np.array_equal(got, exp), "exact solver differs: %s" % {k: c[k] for k in ("w", "h", "ch", "roi", "radius")}
    if c["use_conf"]:
        assert np.array_equal(f.getConfidenceMap(), exp_conf)
    f.setSolver(adf.SOLVER_WAVE)
    got2 = f.filter(c["dl"], c["view"], None, c["dr"] if c["use_conf"] else None, c["roi"])
    if c["use_conf"]:
        assert np.array_equal(f.getConfidenceMap(), exp_conf)
    d = np.abs(got2.astype(np.int64) - exp)
    assert d.max() <= 1 and d.mean() <= 1 / 256, (d.max(), d.mean(), {k: c[k] for k in ("w", "h", "roi", "lam", "sigma")})


# Badly conditioned draws from the same generator (found by running it for 400 seeds): nearly edge-blind
# smoothing (large sigma_color), lambda far above the reference's 8000 or no attenuation between the
# iterations.  There the float32 recurrences themselves are 0.01-0.1 LSB (mean) away from the float64
# solution in EITHER evaluation order (tests/test_gpu_wave.py compares both with oracle/banded_f64.py), so
# the scalar order and the wave solver round differently in up to ~10 % of the pixels -- never by more than
# 1 LSB.  The hard bound is what the port guarantees everywhere; the reference's L1 bar (N/256,
# test_disparity_wls_filter.cpp:105) holds on the reference's own parameter range (BASELINE configs:
# test_gpu_wave.py) but not for arbitrary parameters.
@pytest.mark.parametrize("seed", [152, 199, 229, 291, 354])
def test_badly_conditioned_cases_stay_within_one_lsb(adf, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    c = _case(rng)
    p = oracle.default_params(threads=8, use_confidence=int(c["use_conf"]), sigma_color=c["sigma"],
                              disc_radius=c["radius"], lrc_thresh=c["thresh"], num_iter=c["num_iter"],
                              lambda_attenuation=c["atten"])
    p.lambda_ = c["lam"]
    exp, exp_conf = oracle.wls_filter(c["dl"], c["view"], c["dr"] if c["use_conf"] else None, c["roi"], p)
    f = adf.createDisparityWLSFilterGeneric(c["use_conf"])
    f.setLambda(c["lam"]); f.setSigmaColor(c["sigma"]); f.setDepthDiscontinuityRadius(c["radius"])
    f.setLRCthresh(c["thresh"]); f.setFGSParams(c["atten"], c["num_iter"])
    f.setSolver(adf.SOLVER_EXACT)
    assert np.array_equal(f.filter(c["dl"], c["view"], None, c["dr"] if c["use_conf"] else None, c["roi"]), exp)
    f.setSolver(adf.SOLVER_WAVE)
    got = f.filter(c["dl"], c["view"], None, c["dr"] if c["use_conf"] else None, c["roi"])
    if c["use_conf"]:
        assert np.array_equal(f.getConfidenceMap(), exp_conf)
    d = np.abs(got.astype(np.int64) - exp)
    assert d.max() <= 1 and d.mean() <= 0.15, (d.max(), d.mean())
