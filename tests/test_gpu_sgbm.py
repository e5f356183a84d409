"""GPU tests of the device semi-global matcher (csrc/sgbm_matcher.hip) through the C-ABI.

Integer work: bit-exact against oracle/adf_oracle_sgbm.c (which tests/test_oracle_sgbm.py checks against a direct numpy
statement).  The reference-held anchor is its stereo module's semi-global test (Tsukuba pair + ground truth, <= 10 %)."""
import numpy as np
import pytest

from test_oracle_bm import load_tsukuba
from test_oracle_sgbm import _pair, ref_error_level

pytestmark = pytest.mark.gpu


def _sgbm(adf, nd, bs, md=0, P1=None, P2=None, cap=63, ur=0, mode=2):
    m = adf.StereoSGBM.create(md, nd, bs)
    m.setP1(24 * bs * bs if P1 is None else P1); m.setP2(96 * bs * bs if P2 is None else P2)
    m.setPreFilterCap(cap); m.setUniquenessRatio(ur); m.setMode(mode)
    m.setDisp12MaxDiff(1000000); m.setSpeckleWindowSize(0)          # what createDisparityWLSFilter sets (DF.cpp:389-390)
    return m


def _exp(oracle, a, b, nd, bs, md=0, P1=None, P2=None, cap=63, ur=0):
    return oracle.sgbm_compute(a, b, nd, bs, md, 24 * bs * bs if P1 is None else P1, 96 * bs * bs if P2 is None else P2, cap, ur)


@pytest.mark.parametrize("bs", [1, 3, 5, 7, 9, 11])
def test_every_block_size_bit_exact(adf, oracle, bs):
    a, b = _pair(bs, 41, 150, shift=5)
    assert np.array_equal(_sgbm(adf, 32, bs).compute(a, b), _exp(oracle, a, b, 32, bs))


@pytest.mark.parametrize("nd", [16, 32, 48, 64, 80, 128, 160, 256, 272, 512])
def test_every_disparity_count_bit_exact(adf, oracle, nd):
    """1, 2, 4 and 8 path costs per lane; counts that leave lanes idle (48, 80, 160, 272)."""
    a, b = _pair(nd, 17, nd + 70, shift=9)
    assert np.array_equal(_sgbm(adf, nd, 3).compute(a, b), _exp(oracle, a, b, nd, 3))


@pytest.mark.parametrize("H,W,nd,bs,md,P1,P2,cap,ur,cn", [
    (23, 90, 16, 3, 0, 0, 0, 0, 0, 1),            # every default of cv::StereoSGBM (P1 2, P2 5, cap 15)
    (30, 120, 32, 5, -31, 200, 800, 63, 0, 1),    # the right-view matcher's range (DF.cpp:435)
    (19, 140, 48, 3, 7, 72, 288, 31, 15, 1),      # positive minimum disparity, uniqueness test on
    (25, 100, 32, 7, -12, 10, 100, 20, 5, 1),     # range straddling zero
    (21, 110, 32, 3, 0, 216, 864, 63, 0, 3),      # 3-channel views, the sample's P1 / P2 (24*w*w, 96*w*w)
    (16, 80, 64, 9, -63, 100, 101, 63, 0, 3),     # 3 channels, right matcher's range, P2 = P1 + 1
    (9, 40, 64, 3, 0, 72, 288, 63, 0, 1),         # search range wider than the image: everything invalid
    (3, 70, 16, 11, 0, 72, 288, 63, 0, 1),        # window taller than the image
    (200, 64, 16, 3, 0, 72, 288, 63, 0, 1),       # more rows than one cost band
])
def test_parameter_corners_bit_exact(adf, oracle, H, W, nd, bs, md, P1, P2, cap, ur, cn):
    a, b = _pair(H * W + nd, H, W, cn, shift=6)
    got = _sgbm(adf, nd, bs, md, P1, P2, cap, ur).compute(a, b)
    exp = oracle.sgbm_compute(a, b, nd, bs, md, P1, P2, cap, ur)
    assert got.shape == (H, W) and np.array_equal(got, exp)
    if ur >= 15:                                            # the uniqueness test rejects something (seen before the median)
        raw = oracle.sgbm_compute(a, b, nd, bs, md, P1, P2, cap, ur, want_raw=True)[1]
        assert (raw[:, max(md + nd, 0):W + min(md, 0)] == (md - 1) * 16).any()


def test_reference_fixture_bar_and_parity(adf, oracle):
    left, right, gt = load_tsukuba()
    for bs, P1, P2, ur in ((9, 10, 100, 1), (3, 216, 864, 0)):          # the reference test's penalties; the sample's
        got = _sgbm(adf, 16, bs, 0, P1, P2, 63, ur).compute(left, right)
        assert np.array_equal(got, oracle.sgbm_compute(left, right, 16, bs, 0, P1, P2, 63, ur))
        assert ref_error_level(gt, got) <= 10.0                          # test_block_matching.cpp:231


def test_device_batch_strided_and_color(adf, oracle):
    import torch
    N, H, W = 3, 40, 130
    pairs = [_pair(200 + i, H, W, 3, shift=3 + i) for i in range(N)]
    bl = torch.zeros((N, H, W + 5, 3), dtype=torch.uint8, device="cuda"); br = torch.zeros_like(bl)
    for i, (l, r) in enumerate(pairs):
        bl[i, :, :W] = torch.from_numpy(l).cuda(); br[i, :, :W] = torch.from_numpy(r).cuda()
    out = torch.full((N, H, W + 3), 777, dtype=torch.int16, device="cuda")
    m = _sgbm(adf, 32, 3)
    res = m.compute(bl[:, :, :W], br[:, :, :W], out[:, :, :W])
    torch.cuda.synchronize()
    assert res.data_ptr() == out.data_ptr() and (out[:, :, W:] == 777).all()
    for i, (l, r) in enumerate(pairs):
        assert np.array_equal(out[i, :, :W].cpu().numpy(), _exp(oracle, l, r, 32, 3))
    one = m.compute(bl[1, :, :W], br[1, :, :W])                           # unbatched colour pair
    assert torch.equal(one, out[1, :, :W])


def test_views_to_filtered_disparity_with_the_semi_global_matcher(adf, oracle):
    """The sample's default pipeline (disparity_filtering.cpp:164-189): StereoSGBM in MODE_SGBM_3WAY for both views,
    then the WLS filter set up from the matcher (ROI offsets and radius of the SGBM branch, DF.cpp:404-409), every stage
    on the device, against the same pipeline of the oracle; the filter must not hurt the error against ground truth."""
    import torch
    left, right, gt = load_tsukuba()
    nd, bs = 16, 3
    lm = adf.StereoSGBM.create(0, nd, bs)
    lm.setP1(24 * bs * bs); lm.setP2(96 * bs * bs); lm.setPreFilterCap(63); lm.setMode(adf.StereoSGBM.MODE_SGBM_3WAY)
    wls = adf.createDisparityWLSFilter(lm)
    rm = adf.createRightMatcher(lm)
    assert (lm.getDisp12MaxDiff(), lm.getUniquenessRatio(), rm.getMinDisparity(), rm.getMode()) == (1000000, 0, -nd + 1, 2)
    assert wls.getDepthDiscontinuityRadius() == 2                        # ceil(0.5 * 3), DF.cpp:408
    wls.setLambda(8000.0); wls.setSigmaColor(1.5); wls.setSolver(adf.SOLVER_EXACT)
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    dl = lm.compute(tl, tr); dr = rm.compute(tr, tl)
    out = wls.filter(dl, tl, None, dr)
    torch.cuda.synchronize()
    edl = oracle.sgbm_compute(left, right, nd, bs, 0, 24 * bs * bs, 96 * bs * bs, 63, 0)
    edr = oracle.sgbm_compute(right, left, nd, bs, -nd + 1, 24 * bs * bs, 96 * bs * bs, 63, 0)
    assert np.array_equal(dl.cpu().numpy(), edl) and np.array_equal(dr.cpu().numpy(), edr)
    roi = wls.getROI()
    assert roi == (nd, 0, left.shape[1] - nd, left.shape[0])             # DF.cpp:407
    p = oracle.default_params(threads=4, use_confidence=1, disc_radius=2)
    p.lambda_ = 8000.0; p.sigma_color = 1.5
    exp, exp_conf = oracle.wls_filter(edl, left, edr, roi, p)
    assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(wls.getConfidenceMap().cpu().numpy(), exp_conf)
    x, y, w, h = roi
    g = gt[y:y + h, x:x + w].astype(np.int64); known = g != 0
    mse = lambda d: (((g - d[y:y + h, x:x + w].astype(np.int64))[known]) ** 2).mean() / 256.0
    assert mse(out.cpu().numpy()) <= mse(edl)


def test_matcher_errors(adf):
    a, b = _pair(3, 30, 80)
    for nd, bs in ((24, 3), (16, 4), (16, 13), (528, 3)):
        with pytest.raises(adf.AdfError):
            _sgbm(adf, nd, bs).compute(a, b)
    m = _sgbm(adf, 16, 3); m.setMode(5)
    with pytest.raises(adf.AdfError):
        m.compute(a, b)
    m = _sgbm(adf, 16, 3); m.setSpeckleWindowSize(100)
    with pytest.raises(adf.AdfError):
        m.compute(a, b)
    with pytest.raises(adf.AdfError):
        _sgbm(adf, 16, 3).compute(a, b[:, :-1])


def test_random_parameters_bit_exact(adf, oracle):
    """Seeded fuzz over sizes and every parameter of the matcher (guards the kernels' template / lane arithmetic)."""
    rng = np.random.default_rng(77)
    for case in range(30):
        bs = int(rng.choice([1, 3, 5, 7, 9, 11]))
        nd = 16 * int(rng.integers(1, 13))
        md = int(rng.integers(-nd - 6, 20))
        cn = int(rng.choice([1, 1, 3]))
        H = int(rng.integers(2, 60)); W = int(rng.integers(max(8, nd // 2), nd + 260))
        P1 = int(rng.choice([0, 8, 72, 216, 600])); P2 = int(rng.choice([0, 32, 288, 864, 2400]))
        cap = int(rng.choice([0, 15, 31, 63])); ur = int(rng.choice([0, 0, 5, 15, 40]))
        a, b = _pair(3000 + case, H, W, cn, shift=int(rng.integers(0, 14)))
        got = _sgbm(adf, nd, bs, md, P1, P2, cap, ur).compute(a, b)
        exp = oracle.sgbm_compute(a, b, nd, bs, md, P1, P2, cap, ur)
        assert np.array_equal(got, exp), (case, H, W, cn, nd, bs, md, P1, P2, cap, ur)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("H,W,nd,bs,md,P1,P2,cap,ur,cn", [
    (40, 120, 32, 3, 0, 216, 864, 63, 0, 1),       # wider than tall: diagonals leave through the bottom
    (150, 70, 16, 5, -15, 10, 100, 31, 10, 1),     # taller than the matchable area is wide: they leave through the sides
    (33, 90, 160, 3, 0, 72, 288, 63, 0, 1),        # narrower matchable area than disparities; idle lanes
    (27, 100, 48, 7, -20, 50, 200, 40, 0, 3),      # 3 channels, range straddling zero
    (1, 60, 16, 3, 0, 72, 288, 63, 0, 1),          # a single row: every vertical / diagonal path is one pixel long
    (50, 17, 16, 1, 0, 0, 0, 0, 0, 1),             # a single matchable column
])
def test_five_and_eight_path_modes_bit_exact(adf, oracle, mode, H, W, nd, bs, md, P1, P2, cap, ur, cn):
    """cv::StereoSGBM's default MODE_SGBM (left, up-left, up, up-right, right) and MODE_HH (all eight directions)."""
    a, b = _pair(H * W + nd + mode, H, W, cn, shift=5)
    got = _sgbm(adf, nd, bs, md, P1, P2, cap, ur, mode).compute(a, b)
    exp = oracle.sgbm_compute(a, b, nd, bs, md, P1, P2, cap, ur, mode=mode)
    assert np.array_equal(got, exp)


def test_modes_on_the_reference_fixture(adf, oracle):
    left, right, gt = load_tsukuba()
    for mode in (0, 1, 2):
        got = _sgbm(adf, 16, 3, 0, 216, 864, 63, 0, mode).compute(left, right)
        assert np.array_equal(got, oracle.sgbm_compute(left, right, 16, 3, 0, 216, 864, 63, 0, mode=mode))
        assert ref_error_level(gt, got) <= 10.0


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("md,disp12,nd", [(0, 0, 16), (0, 1, 32), (0, 3, 16), (-15, 1, 16), (4, 2, 16), (0, 1, 160)])
def test_matchers_own_left_right_check(adf, oracle, mode, md, disp12, nd):
    """disp12MaxDiff inside the matcher: cv::StereoSGBM::create's default leaves it on (0 -> 1)."""
    rng = np.random.default_rng(400 + mode + nd)
    H, W = 30, nd + 110
    base = rng.integers(0, 256, (H, W + 60), dtype=np.uint8)
    a = np.ascontiguousarray(base[:, 20:20 + W])
    b = np.ascontiguousarray(base[:, 23:23 + W]).copy()
    b[:, W // 2:] = base[:, 20 + 9 + W // 2:20 + 9 + W]              # right half at a larger disparity: occlusions
    m = _sgbm(adf, nd, 3, md, 72, 288, 63, 0, mode); m.setDisp12MaxDiff(disp12)
    got = m.compute(a, b)
    exp = oracle.sgbm_compute(a, b, nd, 3, md, 72, 288, 63, 0, mode=mode, disp12_max_diff=disp12)
    assert np.array_equal(got, exp)


def test_create_defaults_run(adf, oracle):
    """StereoSGBM.create(minDisparity, numDisparities, blockSize) with everything else at cv::StereoSGBM's defaults:
    MODE_SGBM, P1 = P2 = 0 (-> 2 / 5), preFilterCap 0 (-> 15), uniquenessRatio 0, disp12MaxDiff 0 (-> 1, check on)."""
    a, b = _pair(12, 40, 140, shift=7)
    m = adf.StereoSGBM.create(0, 32, 5)
    assert np.array_equal(m.compute(a, b), oracle.sgbm_compute(a, b, 32, 5, 0, 0, 0, 0, 0, mode=0, disp12_max_diff=0))


def test_host_batch_with_row_padding(adf, oracle):
    """adf_sgbm_compute_host on a batch of numpy pairs whose rows carry padding (strides larger than the width)."""
    N, H, W = 2, 31, 120
    pairs = [_pair(900 + i, H, W, 1, shift=4 + i) for i in range(N)]
    bl = np.zeros((N, H, W + 9), np.uint8); br = np.zeros((N, H, W + 5), np.uint8)
    for i, (l, r) in enumerate(pairs):
        bl[i, :, :W] = l; br[i, :, :W] = r
    out = np.full((N, H, W + 3), 555, np.int16)
    _sgbm(adf, 32, 5).compute(bl[:, :, :W], br[:, :, :W], out[:, :, :W])
    assert (out[:, :, W:] == 555).all()
    for i, (l, r) in enumerate(pairs):
        assert np.array_equal(out[i, :, :W], _exp(oracle, l, r, 32, 5))
