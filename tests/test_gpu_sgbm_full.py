"""The semi-global matcher at the sizes BASELINE.json names (ADVICE r2 test_gpu_sgbm.py:32).

tests/test_gpu_sgbm.py covers every template / parameter corner on small images; here the SAME kernels run at
config 1 (640x480 / 64 disparities), config 2 (1920x1080 / 160) and config 3 / 4 (3840x2160 / 256: two 3.96 GB
volumes per image, a second pair's volumes start past 2^32 bytes) with the sample's matcher settings
(samples/disparity_filtering.cpp:166-176: MODE_SGBM_3WAY, P1 = 24 w^2, P2 = 96 w^2, preFilterCap 63), both views,
bit for bit against oracle/adf_oracle_sgbm.c, and their maps go through the filter set up from the matcher
(disparity_filters.cpp:404-409, 432-445) against the oracle's pipeline.  Real image content: the reference's own
KITTI-shaped pair (modules/stereo/testdata, data fixtures under tests/golden/) tiled to the size."""
import numpy as np
import pytest

from test_gpu_real_guides import MAX_DIF, MAX_MEAN_DIF, load_kitti

pytestmark = pytest.mark.gpu


def tiled_pair(H, W, roll=0):
    """Left and right tiled the same way, so disparities stay consistent inside each tile."""
    left, right = load_kitti()
    ty, tx = -(-H // left.shape[0]), -(-W // left.shape[1])
    L = np.tile(left, (ty, tx))[:H, :W]; R = np.tile(right, (ty, tx))[:H, :W]
    if roll:
        L = np.roll(L, roll, 0); R = np.roll(R, roll, 0)
    return np.ascontiguousarray(L), np.ascontiguousarray(R)


def sample_matcher(adf, nd, bs=3):
    lm = adf.StereoSGBM.create(0, nd, bs)
    lm.setP1(24 * bs * bs); lm.setP2(96 * bs * bs); lm.setPreFilterCap(63); lm.setMode(adf.StereoSGBM.MODE_SGBM_3WAY)
    return lm


def oracle_maps(oracle, L, R, nd, bs=3):
    edl = oracle.sgbm_compute(L, R, nd, bs, 0, 24 * bs * bs, 96 * bs * bs, 63, 0)
    edr = oracle.sgbm_compute(R, L, nd, bs, -nd + 1, 24 * bs * bs, 96 * bs * bs, 63, 0)      # DF.cpp:435
    return edl, edr


def sgbm_then_filter(adf, oracle, H, W, nd, guide_channels):
    import torch
    L, R = tiled_pair(H, W)
    lm = sample_matcher(adf, nd)
    wls = adf.createDisparityWLSFilter(lm)                   # DF.cpp:404-409: ROI from the matcher, radius ceil(0.5 * 3)
    rm = adf.createRightMatcher(lm)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    dl = lm.compute(tl, tr); dr = rm.compute(tr, tl)
    torch.cuda.synchronize()
    edl, edr = oracle_maps(oracle, L, R, nd)
    assert np.array_equal(dl.cpu().numpy(), edl), "left-view matcher differs from its oracle"
    assert np.array_equal(dr.cpu().numpy(), edr), "right-view matcher differs from its oracle"
    assert (edl[:, nd:] >= 0).mean() > 0.8                   # the pair really matches: not a map of invalid values
    guide = L if guide_channels == 1 else np.ascontiguousarray(np.stack([L, np.roll(L, 1, 0), np.roll(L, 1, 1)], axis=2))
    tg = torch.from_numpy(guide).cuda()
    wls.setLambda(8000.0); wls.setSigmaColor(1.5)
    p = oracle.default_params(threads=8, use_confidence=1, disc_radius=2, sigma_color=1.5)
    p.lambda_ = 8000.0
    roi = (nd, 0, W - nd, H)
    exp, exp_conf = oracle.wls_filter(edl, guide, edr, roi, p)
    for solver in (adf.SOLVER_EXACT, adf.SOLVER_WAVE):
        wls.setSolver(solver)
        out = wls.filter(dl, tg, None, dr)
        torch.cuda.synchronize()
        assert wls.getROI() == roi and wls.getLastSolver() == solver
        assert np.array_equal(wls.getConfidenceMap().cpu().numpy(), exp_conf)
        d = np.abs(out.cpu().numpy().astype(np.int64) - exp.astype(np.int64))
        if solver == adf.SOLVER_EXACT:
            assert d.max() == 0
        else:
            assert d.max() <= MAX_DIF and d.mean() <= MAX_MEAN_DIF, (d.max(), d.mean())
    assert (exp_conf > 0).mean() > 0.2 and (exp[:, nd:] != edl[:, nd:]).mean() > 0.3      # the filter had work to do


def test_config1_sgbm_pair_through_the_filter(adf, oracle):
    """BASELINE config 1: single 640x480 StereoSGBM pair, 64 disparities => ROI (64,0,576,480)."""
    sgbm_then_filter(adf, oracle, 480, 640, 64, 3)


def test_config2_sgbm_pair_through_the_filter(adf, oracle):
    """BASELINE config 2: single 1920x1080 SGBM disparity, 160 disparities => ROI (160,0,1760,1080)."""
    sgbm_then_filter(adf, oracle, 1080, 1920, 160, 3)


def test_4k_256_two_pair_batch_beyond_4gb_volumes(adf, oracle):
    """Config 3 / 4's pair: 3840x2160, 256 disparities.  One image's C and S volumes are 3.96 GB each, so in a two-pair
    call the second pair's volumes start past 2^32 bytes.  Pair 0 of the batch against the oracle (one ~40 s oracle
    call), pair 1 (different content) against the single-pair call."""
    import torch
    H, W, nd = 2160, 3840, 256
    L0, R0 = tiled_pair(H, W)
    L1, R1 = tiled_pair(H, W, roll=101)
    lm = sample_matcher(adf, nd)
    lm.setDisp12MaxDiff(1000000)
    tl = torch.from_numpy(np.stack([L0, L1])).cuda(); tr = torch.from_numpy(np.stack([R0, R1])).cuda()
    both = lm.compute(tl, tr)
    torch.cuda.synchronize()
    one = lm.compute(tl[1], tr[1])
    torch.cuda.synchronize()
    assert torch.equal(both[1], one), "pair 1 of the batch differs from the single-pair call"
    assert not torch.equal(both[0], both[1])
    exp = oracle.sgbm_compute(L0, R0, nd, 3, 0, 216, 864, 63, 0)
    assert np.array_equal(both[0].cpu().numpy(), exp)
    assert (exp[:, nd:] >= 0).mean() > 0.8


def test_wide_strip_left_right_check_in_chunks(adf, oracle, monkeypatch):
    """3840 columns, 256 disparities, the matcher's own left-right check on (its reverse map lives in LDS: more than
    48 KB per workgroup, the hipFuncSetAttribute branch), a 3-pair batch under a workspace limit that forces two
    chunks (ADF_WS_LIMIT_GB is read when the matcher is created)."""
    import torch
    H, W, nd = 96, 3840, 256
    pairs = [tiled_pair(H, W, roll=17 * i) for i in range(3)]
    monkeypatch.setenv("ADF_WS_LIMIT_GB", "0.8")             # one image in flight needs 0.36 GB: chunks of 2 + 1
    m = adf.StereoSGBM.create(0, nd, 3)
    m.setP1(216); m.setP2(864); m.setPreFilterCap(63); m.setMode(adf.StereoSGBM.MODE_SGBM_3WAY)
    m.setDisp12MaxDiff(1)
    tl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda(); tr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    got = m.compute(tl, tr).cpu().numpy()
    n_rejected = 0
    for i, (L, R) in enumerate(pairs):
        exp = oracle.sgbm_compute(L, R, nd, 3, 0, 216, 864, 63, 0, mode=2, disp12_max_diff=1)
        assert np.array_equal(got[i], exp), i
        off = oracle.sgbm_compute(L, R, nd, 3, 0, 216, 864, 63, 0, mode=2)
        n_rejected += int((exp != off).sum())
    assert n_rejected > 0                                    # the check really rejected something
