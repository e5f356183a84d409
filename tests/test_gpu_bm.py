"""GPU tests of the device block matcher (csrc/bm_matcher.hip) through the C-ABI.

Integer work: bit-exact against oracle/adf_oracle_bm.c.  The reference-held anchor is its stereo module's
block-matching test (Tsukuba pair + ground truth, <= 20 % bad pixels; see tests/test_oracle_bm.py)."""
import numpy as np
import pytest

from test_oracle_bm import error_level, load_tsukuba

pytestmark = pytest.mark.gpu


def _views(seed, H, W, shift=5):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H, W + 96), dtype=np.uint8)
    base = (base // 2 + np.roll(base, 1, 1) // 4 + np.roll(base, 1, 0) // 4).astype(np.uint8)
    base[H // 4:H // 2, 60:60 + W // 4] = 90                                        # a textureless patch
    base[H // 2:, 50:50 + W // 3] = (40 + 150 * ((np.arange(W // 3) // 3) % 2)).astype(np.uint8)   # a repetitive one
    return np.ascontiguousarray(base[:, 40:40 + W]), np.ascontiguousarray(np.roll(base, -shift, 1)[:, 40:40 + W])


def _bm(adf, nd, wsz, md=0, cap=31, texthr=0, uniq=0):
    bm = adf.StereoBM.create(nd, wsz)
    bm.setMinDisparity(md); bm.setPreFilterCap(cap); bm.setTextureThreshold(texthr); bm.setUniquenessRatio(uniq)
    return bm


@pytest.mark.parametrize("wsz", [5, 7, 9, 11, 13, 15, 17, 19, 21])
def test_every_block_size_bit_exact(adf, oracle, wsz):
    left, right = _views(wsz, 61, 203)
    got = _bm(adf, 32, wsz).compute(left, right)
    assert np.array_equal(got, oracle.bm_compute(left, right, 32, wsz))


@pytest.mark.parametrize("H,W,nd,wsz,md,cap,texthr,uniq", [
    (23, 64, 16, 5, 0, 31, 0, 0),          # narrower than one wave tile
    (40, 130, 16, 9, 0, 31, 10, 15),       # cv::StereoBM's default tests
    (37, 211, 48, 7, -47, 31, 0, 0),       # the right-view matcher's range (DF.cpp:424)
    (66, 180, 32, 15, 3, 63, 300, 5),      # positive minimum disparity, texture and uniqueness rejections
    (41, 150, 64, 11, -20, 15, 0, 30),     # range straddling zero
    (22, 300, 128, 21, 0, 1, 0, 0),        # widest window, smallest cap
    (9, 90, 16, 7, 0, 31, 0, 0),           # fewer rows than two row groups
    (30, 200, 32, 21, 0, 31, 5, 20),       # uniqueness instantiation (two columns per lane), widest window
    (25, 150, 16, 5, -3, 31, 0, 10),       # ... narrowest window
    (33, 700, 48, 13, 0, 31, 0, 12),       # ... several column tiles
])
def test_parameter_corners_bit_exact(adf, oracle, H, W, nd, wsz, md, cap, texthr, uniq):
    left, right = _views(H * W, H, W, shift=7)
    got = _bm(adf, nd, wsz, md, cap, texthr, uniq).compute(left, right)
    exp = oracle.bm_compute(left, right, nd, wsz, md, cap, texthr, uniq)
    assert np.array_equal(got, exp)
    if texthr or uniq:
        assert (exp[:, max(md + nd - 1, 0) + wsz // 2:W - max(-md, 0) - wsz // 2] == (md - 1) * 16).any()   # the tests reject something


def test_search_range_wider_than_image_is_all_rejected(adf, oracle):
    left, right = _views(1, 30, 60)
    got = _bm(adf, 64, 9).compute(left, right)
    assert (got == -16).all() and np.array_equal(got, oracle.bm_compute(left, right, 64, 9))


def test_reference_fixture_bar_and_parity(adf, oracle):
    """Tsukuba pair from the reference's stereo test data: bit-exact against the oracle and within the
    reference test's own accuracy bar (test_block_matching.cpp:148)."""
    left, right, gt = load_tsukuba()
    for wsz in (9, 15):
        got = _bm(adf, 16, wsz).compute(left, right)
        assert np.array_equal(got, oracle.bm_compute(left, right, 16, wsz))
        assert error_level(gt, got) <= 20.0


def test_device_batch_and_strided_views(adf, oracle):
    import torch
    N, H, W = 3, 50, 170
    pairs = [_views(100 + i, H, W, shift=3 + i) for i in range(N)]
    big_l = torch.zeros((N, H, W + 13), dtype=torch.uint8, device="cuda")
    big_r = torch.zeros((N, H, W + 13), dtype=torch.uint8, device="cuda")
    for i, (l, r) in enumerate(pairs):
        big_l[i, :, :W] = torch.from_numpy(l).cuda(); big_r[i, :, :W] = torch.from_numpy(r).cuda()
    out = torch.full((N, H, W + 6), 777, dtype=torch.int16, device="cuda")
    bm = _bm(adf, 32, 9)
    res = bm.compute(big_l[:, :, :W], big_r[:, :, :W], out[:, :, :W])
    torch.cuda.synchronize()
    assert res.data_ptr() == out.data_ptr()
    assert (out[:, :, W:] == 777).all()                       # nothing written past the row
    for i, (l, r) in enumerate(pairs):
        assert np.array_equal(out[i, :, :W].cpu().numpy(), oracle.bm_compute(l, r, 32, 9))


def test_matcher_errors(adf):
    left, right = _views(2, 40, 80)
    for nd, wsz in ((0 + 24, 9), (16, 8), (16, 3), (16, 23), (16, 41)):
        with pytest.raises(adf.AdfError):
            bm = adf.StereoBM.create(nd, wsz); bm.compute(left, right)
    with pytest.raises(adf.AdfError):
        _bm(adf, 16, 9, cap=64).compute(left, right)
    with pytest.raises(adf.AdfError):
        _bm(adf, 16, 9).compute(left, right[:, :-1])
    bm = _bm(adf, 16, 9); bm.setSpeckleWindowSize(100)
    with pytest.raises(adf.AdfError):
        bm.compute(left, right)
    with pytest.raises(adf.AdfError):                       # both-views extension: device tensors only
        _bm(adf, 16, 9).computeBoth(left, right)


def test_views_to_filtered_disparity_on_device(adf, oracle):
    """The sample's pipeline (disparity_filtering.cpp:151-189: left matcher, right matcher, wls filter) with
    every stage on the device, against the same pipeline of the oracle; and the filter must not hurt the
    error against the fixture's ground truth."""
    import torch
    left, right, gt = load_tsukuba()
    nd, wsz = 16, 9
    lm = adf.StereoBM.create(nd, wsz)
    wls = adf.createDisparityWLSFilter(lm)                    # forces textureThreshold = uniquenessRatio = 0
    rm = adf.createRightMatcher(lm)
    assert (lm.getTextureThreshold(), lm.getUniquenessRatio(), rm.getMinDisparity()) == (0, 0, -(0 + nd) + 1)
    wls.setLambda(8000.0); wls.setSigmaColor(1.5); wls.setSolver(adf.SOLVER_EXACT)
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    dl = lm.compute(tl, tr)
    dr = rm.compute(tr, tl)
    out = wls.filter(dl, tl, None, dr)
    torch.cuda.synchronize()
    # oracle pipeline
    edl = oracle.bm_compute(left, right, nd, wsz, 0)
    edr = oracle.bm_compute(right, left, nd, wsz, -(0 + nd) + 1)
    assert np.array_equal(dl.cpu().numpy(), edl) and np.array_equal(dr.cpu().numpy(), edr)
    roi = wls.getROI()
    p = oracle.default_params(threads=4, use_confidence=1, disc_radius=wls.getDepthDiscontinuityRadius())
    p.lambda_ = 8000.0; p.sigma_color = 1.5
    exp, exp_conf = oracle.wls_filter(edl, left, edr, roi, p)
    assert np.array_equal(out.cpu().numpy(), exp)
    assert np.array_equal(wls.getConfidenceMap().cpu().numpy() if hasattr(wls.getConfidenceMap(), "cpu") else wls.getConfidenceMap(), exp_conf)
    x, y, w, h = roi
    g = gt[y:y + h, x:x + w].astype(np.int64); known = g != 0
    mse = lambda d: (((g - d[y:y + h, x:x + w].astype(np.int64))[known]) ** 2).mean() / 256.0
    assert mse(out.cpu().numpy()) <= mse(edl)


def test_random_parameters_bit_exact(adf, oracle):
    """Seeded fuzz over sizes and every parameter of the matcher (the reference has no such test for a matcher:
    this guards the kernel's template / tiling arithmetic)."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        wsz = int(rng.choice([5, 7, 9, 11, 13, 15, 17, 19, 21]))
        nd = 16 * int(rng.integers(1, 7))
        md = int(rng.integers(-nd - 10, 24))
        H = int(rng.integers(wsz + 1, 70)); W = int(rng.integers(max(wsz + 1, 40), 560))
        cap = int(rng.integers(1, 64)); texthr = int(rng.choice([0, 0, 10, 200])); uniq = int(rng.choice([0, 0, 5, 25]))
        left, right = _views(1000 + case, H, W, shift=int(rng.integers(0, 12)))
        got = _bm(adf, nd, wsz, md, cap, texthr, uniq).compute(left, right)
        exp = oracle.bm_compute(left, right, nd, wsz, md, cap, texthr, uniq)
        assert np.array_equal(got, exp), (case, H, W, nd, wsz, md, cap, texthr, uniq)


def test_host_batch_with_row_padding(adf, oracle):
    """adf_bm_compute_host on a batch of numpy pairs whose rows carry padding (strides larger than the width)."""
    N, H, W = 2, 37, 150
    pairs = [_views(300 + i, H, W, shift=4 + i) for i in range(N)]
    bl = np.zeros((N, H, W + 9), np.uint8); br = np.zeros((N, H, W + 5), np.uint8)
    for i, (l, r) in enumerate(pairs):
        bl[i, :, :W] = l; br[i, :, :W] = r
    out = np.full((N, H, W + 3), 555, np.int16)
    _bm(adf, 32, 11, 0, 31, 10, 15).compute(bl[:, :, :W], br[:, :, :W], out[:, :, :W])
    assert (out[:, :, W:] == 555).all()
    for i, (l, r) in enumerate(pairs):
        assert np.array_equal(out[i, :, :W], oracle.bm_compute(l, r, 32, 11, 0, 31, 10, 15))


@pytest.mark.parametrize("nd,wsz,md,texthr,uniq", [(32, 9, 0, 0, 0), (48, 15, 5, 0, 0), (16, 7, -4, 10, 15), (64, 21, 0, 0, 8)])
def test_both_views_in_one_launch(adf, oracle, nd, wsz, md, texthr, uniq):
    """adf_bm_compute_both_device (extension): identical to the left matcher and createRightMatcher's matcher run
    separately (DF.cpp:417-431: views swapped, minDisparity = -(min+num)+1, texture / uniqueness tests off)."""
    import torch
    N, H, W = 2, 45, 330
    pairs = [_views(500 + i + nd, H, W, shift=3 + 2 * i) for i in range(N)]
    tl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    tr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    lm = _bm(adf, nd, wsz, md, 31, texthr, uniq)
    dl, dr = lm.computeBoth(tl, tr)
    torch.cuda.synchronize()
    for i, (l, r) in enumerate(pairs):
        assert np.array_equal(dl[i].cpu().numpy(), oracle.bm_compute(l, r, nd, wsz, md, 31, texthr, uniq))
        assert np.array_equal(dr[i].cpu().numpy(), oracle.bm_compute(r, l, nd, wsz, -(md + nd) + 1, 31, 0, 0))
    rm = adf.createRightMatcher(lm)
    assert torch.equal(dr, rm.compute(tr, tl)) and torch.equal(dl, lm.compute(tl, tr))
    one_l, one_r = lm.computeBoth(tl[0], tr[0])                      # unbatched
    assert torch.equal(one_l, dl[0]) and torch.equal(one_r, dr[0])


@pytest.mark.parametrize("H,W,nd,wsz,md,uniq", [(40, 7680, 512, 15, 0, 0), (36, 5000, 1024, 9, -300, 7), (33, 4099, 496, 21, 16, 0)])
def test_wide_images_and_long_searches(adf, oracle, H, W, nd, wsz, md, uniq):
    """Many column tiles and searches far longer than a tile (8K-wide rows, up to 1024 disparities)."""
    import torch
    left, right = _views(H + W, H, W, shift=37)
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    bm = _bm(adf, nd, wsz, md, 31, 0, uniq)
    exp = oracle.bm_compute(left, right, nd, wsz, md, 31, 0, uniq)
    assert np.array_equal(bm.compute(tl, tr).cpu().numpy(), exp)
    if uniq == 0:
        dl, dr = bm.computeBoth(tl, tr)
        assert np.array_equal(dl.cpu().numpy(), exp)
        assert np.array_equal(dr.cpu().numpy(), oracle.bm_compute(right, left, nd, wsz, -(md + nd) + 1, 31, 0, 0))
