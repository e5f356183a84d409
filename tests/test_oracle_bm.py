"""CPU tests of the block-matcher oracle (oracle/adf_oracle_bm.c).

cv::StereoBM is external to the reference (parity unpinned at the calib3d boundary); what the reference does
hold is its stereo module's block-matching test: the Tsukuba pair testdata/imL2l.bmp (left) / imL2.bmp
(right) against testdata/groundtruth.bmp (disparity * 16), at most 20 % of all pixels with a known ground
truth off by more than 2 * 16 (modules/stereo/test/test_block_matching.cpp:61-82, 88-92, 148).  The three
files are copied as data under tests/golden/."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_tsukuba():
    from PIL import Image
    imgs = []
    for n in ("stereo_left.bmp", "stereo_right.bmp", "stereo_groundtruth.bmp"):
        p = os.path.join(GOLDEN, n)
        if not os.path.exists(p):
            pytest.skip("fixture %s not present" % n)
        imgs.append(np.array(Image.open(p).convert("L")))
    return imgs


def error_level(gt, disp16):
    """test_block_matching.cpp:61-82 (the test compares 8-bit maps of disparity*16)."""
    out8 = np.clip(disp16, 0, 255).astype(np.int64)
    bad = (gt != 0) & (np.abs(gt.astype(np.int64) - out8) > 2 * 16)
    return 100.0 * bad.sum() / gt.size


def naive_bm(left, right, nd, wsz, md=0, cap=31, texthr=0, uniq=0):
    """Independent (slow, direct) statement of the same definition, for small images."""
    H, W = left.shape
    w2 = wsz // 2

    def prefilter(img):
        a = img.astype(np.int64)
        rows = np.arange(H)
        up = np.where(rows > 0, rows - 1, 1 if H > 1 else 0)
        dn = np.where(rows < H - 1, rows + 1, H - 2 if H > 1 else 0)
        out = np.full((H, W), cap, np.int64)
        d = lambda r: r[:, 2:] - r[:, :-2]
        out[:, 1:-1] = np.clip(d(a[up]) + 2 * d(a) + d(a[dn]), -cap, cap) + cap
        return out

    L, R = prefilter(left), prefilter(right)
    maxd = md + nd - 1
    xs, xe = max(maxd, 0) + w2, W - max(-md, 0) - w2
    out = np.full((H, W), (md - 1) * 16, np.int64)
    for y in range(w2, H - w2):                               # calib3d's valid rectangle: full windows only
        rows = np.arange(y - w2, y + w2 + 1)
        for x in range(xs, xe):
            lw = L[rows, x - w2:x + w2 + 1]
            S = np.array([np.abs(lw - R[rows, x - w2 - (md + k):x + w2 + 1 - (md + k)]).sum() for k in range(nd)])
            bk = nd - 1 - int(np.argmin(S[::-1]))            # ties: largest disparity
            best = int(S[bk])
            if np.abs(lw - cap).sum() < texthr:
                continue
            if uniq > 0:
                thresh = best + best * uniq // 100
                far = np.abs(np.arange(nd) - bk) > 1
                if (S[far] <= thresh).any():
                    continue
            p = int(S[bk - 1 if bk > 0 else 1]); n = int(S[bk + 1 if bk < nd - 1 else nd - 2])
            dd = p + n - 2 * best + abs(p - n)
            frac = int((p - n) * 256 / dd) if dd != 0 else 0   # C division truncates toward zero
            out[y, x] = ((bk + md) * 256 + frac + 15) >> 4
    return out.astype(np.int16)


@pytest.mark.parametrize("nd,wsz,md,texthr,uniq", [(16, 5, 0, 0, 0), (16, 9, 0, 10, 15), (32, 7, -31, 0, 0),
                                                  (16, 11, 3, 40, 5), (16, 21, -8, 0, 10)])
def test_oracle_matches_direct_statement(oracle, nd, wsz, md, texthr, uniq):
    rng = np.random.default_rng(nd * 100 + wsz)
    H, W = 37, 83
    base = rng.integers(0, 256, (H, W + 64), dtype=np.uint8)
    base = (base // 2 + np.roll(base, 1, 1) // 4 + np.roll(base, 1, 0) // 4).astype(np.uint8)   # some smoothness
    left = base[:, 20:20 + W]
    right = np.roll(base, -5, 1)[:, 20:20 + W]
    got = oracle.bm_compute(left, right, nd, wsz, md, 31, texthr, uniq)
    exp = naive_bm(left, right, nd, wsz, md, 31, texthr, uniq)
    assert np.array_equal(got, exp)
    assert ((got == (md - 1) * 16).mean() < 1.0)


def test_oracle_prefilter_definition(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (9, 13), dtype=np.uint8)
    out = oracle.bm_prefilter_xsobel(img, 31)
    assert (out[:, 0] == 31).all() and (out[:, -1] == 31).all()
    a = img.astype(np.int64)
    y, x = 4, 6
    v = (a[y - 1, x + 1] - a[y - 1, x - 1]) + 2 * (a[y, x + 1] - a[y, x - 1]) + (a[y + 1, x + 1] - a[y + 1, x - 1])
    assert out[y, x] == np.clip(v, -31, 31) + 31
    v0 = 2 * (a[1, x + 1] - a[1, x - 1]) + 2 * (a[0, x + 1] - a[0, x - 1])      # row -1 reflects to row 1
    assert out[0, x] == np.clip(v0, -31, 31) + 31


@pytest.mark.parametrize("wsz", [9, 11, 15])
def test_oracle_meets_reference_block_matching_bar(oracle, wsz):
    """The reference's own known-answer check for a block matcher (test_block_matching.cpp:148: <= 20 %)."""
    left, right, gt = load_tsukuba()
    disp = oracle.bm_compute(left, right, 16, wsz)
    assert error_level(gt, disp) <= 20.0
    # a swapped pair must fail the bar: the check is not vacuous
    assert error_level(gt, oracle.bm_compute(right, left, 16, wsz)) > 20.0


def test_right_matcher_convention(oracle):
    """disparity_filters.cpp:421-431: the right view is matched with minDisparity = -(min+num)+1 and the views
    swapped; its disparities are the negated left ones at the corresponding pixel (what the LRC test of
    disparity_filters.cpp:331-335 relies on)."""
    left, right, gt = load_tsukuba()
    nd, wsz = 16, 9
    dl = oracle.bm_compute(left, right, nd, wsz, 0)
    dr = oracle.bm_compute(right, left, nd, wsz, -(0 + nd) + 1)
    H, W = dl.shape
    ys, xs = np.nonzero(dl >= 0)
    xr = xs - (dl[ys, xs] >> 4)
    ok = (xr >= 0) & (dr[ys, np.clip(xr, 0, W - 1)] > (-(nd) + 1 - 1) * 16)
    agree = np.abs(dl[ys, xs][ok].astype(np.int64) + dr[ys, xr][ok]) < 24          # LRC_thresh default
    assert agree.mean() > 0.7
