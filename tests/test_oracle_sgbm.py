"""CPU tests of the semi-global matcher oracle (oracle/adf_oracle_sgbm.c).

cv::StereoSGBM is external to the reference (parity unpinned at the calib3d boundary); the anchor the reference holds
is its stereo module's semi-global test: the Tsukuba pair against testdata/groundtruth.bmp, 16 disparities, P1 = 10,
P2 = 100, at most 10 % of the pixels off by more than 2*16 after the CV_16S map is scaled to 8 bits by
255/(max-min) (modules/stereo/test/test_block_matching.cpp:157-238).  The three files are data under tests/golden/."""
import numpy as np
import pytest

from test_oracle_bm import load_tsukuba

SHRT_MAX = 32767


def ref_error_level(gt, disp16):
    """test_block_matching.cpp:218-231: convertTo(CV_8U, 255/(max-min)) then errorLevel (:61-82)."""
    d = disp16.astype(np.float64)
    t8 = np.clip(np.rint(d * 255.0 / (d.max() - d.min())), 0, 255).astype(np.int64)
    bad = (gt != 0) & (np.abs(gt.astype(np.int64) - t8) > 2 * 16)
    return 100.0 * bad.sum() / gt.size


DIRS = {2: [(1, 0), (0, 1), (-1, 0)],                                                       # MODE_SGBM_3WAY
        0: [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0)],                                      # MODE_SGBM
        1: [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]}          # MODE_HH


def naive_sgbm(img1, img2, nd, bs, md=0, P1=0, P2=0, cap=0, ur=0, mode=2, disp12=1000000):
    """Independent direct statement (numpy, whole cost volume in memory) of the definition in adf_oracle_sgbm.c."""
    a = img1.astype(np.int64); b = img2.astype(np.int64)
    if a.ndim == 2:
        a = a[:, :, None]; b = b[:, :, None]
    H, W, cn = a.shape
    ftz = max(cap, 15) | 1
    P1 = P1 if P1 > 0 else 2
    P2 = max(P2 if P2 > 0 else 5, P1 + 1)

    def signals(im):
        up = np.concatenate([im[:1], im[:-1]]); dn = np.concatenate([im[1:], im[-1:]])
        sig = np.full((H, W, 2 * cn), ftz, np.int64)
        if W > 2:
            g = (im[:, 2:] - im[:, :-2]) * 2 + up[:, 2:] - up[:, :-2] + dn[:, 2:] - dn[:, :-2]
            sig[:, 1:-1, :cn] = np.clip(g, -ftz, ftz) + ftz
            sig[:, 1:-1, cn:] = im[:, 1:-1]
        left = np.concatenate([sig[:, :1], sig[:, :-1]], 1); right = np.concatenate([sig[:, 1:], sig[:, -1:]], 1)
        vl = (sig + left) // 2; vr = (sig + right) // 2
        vl[:, 0] = sig[:, 0]; vr[:, -1] = sig[:, -1]
        return sig, np.minimum(np.minimum(vl, vr), sig), np.maximum(np.maximum(vl, vr), sig)

    u, u0, u1 = signals(a); v, v0, v1 = signals(b)
    maxd = md + nd
    minx1 = max(maxd, 0); maxx1 = W + min(md, 0); w1 = maxx1 - minx1
    out = np.full((H, W), (md - 1) * 16, np.int64)
    if w1 <= 0:
        return out
    pix = np.zeros((H, w1, nd), np.int64)
    xs = np.arange(minx1, maxx1)
    for k in range(nd):
        x2 = xs - (md + k)
        c0 = np.maximum(np.maximum(0, u[:, xs] - v1[:, x2]), v0[:, x2] - u[:, xs])
        c1 = np.maximum(np.maximum(0, v[:, x2] - u1[:, xs]), u0[:, xs] - v[:, x2])
        m = np.minimum(c0, c1)
        m[:, :, cn:] >>= 2
        pix[:, :, k] = m.sum(2)
    r = bs // 2
    yy = np.clip(np.arange(-r, H + r), 0, H - 1); xx = np.clip(np.arange(-r, w1 + r), 0, w1 - 1)
    Cv = np.zeros_like(pix)
    for dy in range(bs):
        for dx in range(bs):
            Cv += pix[yy[dy:dy + H]][:, xx[dx:dx + w1]]
    Cv = np.minimum(Cv, SHRT_MAX)

    def step(Cp, Lp, mp):
        big = np.array([SHRT_MAX])
        lm = np.concatenate([big, Lp[:-1]]) + P1; lp = np.concatenate([Lp[1:], big]) + P1
        L = Cp + np.minimum(np.minimum(Lp, lm), np.minimum(lp, mp + P2)) - (mp + P2)
        L = np.clip(L, -32768, SHRT_MAX)
        return L, L.min()

    # every path of the mode: L volume by direct recursion on the pixel before (x - dx, y - dy); zeros outside
    S = np.zeros((H, w1, nd), np.int64)
    for dx, dy in DIRS[mode]:
        L = np.zeros((H, w1, nd), np.int64); M = np.zeros((H, w1), np.int64)
        for y in (range(H) if dy >= 0 else range(H - 1, -1, -1)):
            for x in (range(w1) if dx >= 0 else range(w1 - 1, -1, -1)):
                px, py = x - dx, y - dy
                if 0 <= px < w1 and 0 <= py < H:
                    L[y, x], M[y, x] = step(Cv[y, x], L[py, px], M[py, px])
                else:
                    L[y, x], M[y, x] = step(Cv[y, x], np.zeros(nd, np.int64), 0)
        S = np.clip(S + L, -32768, SHRT_MAX)
    maxdiff = disp12 if disp12 > 0 else 1
    for y in range(H):
        d2p = np.full(W, (md - 1) * 16, np.int64); d2c = np.full(W, SHRT_MAX, np.int64)
        for x in range(w1 - 1, -1, -1):                              # from the right (stereo_binary_sgbm.cpp:456)
            Sp = S[y, x]
            best = int(np.argmin(Sp)); ms = int(Sp[best])            # argmin: first minimum
            if ms >= SHRT_MAX:
                continue
            if ur > 0 and any(Sp[d] * (100 - ur) < ms * 100 and abs(best - d) > 1 for d in range(nd)):
                continue
            d = best
            x2 = x + minx1 - d - md
            if d2c[x2] > ms:
                d2c[x2] = ms; d2p[x2] = d + md
            if 0 < d < nd - 1:
                den = max(int(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d]), 1)
                num = int(Sp[d - 1] - Sp[d + 1]) * 16 + den
                d = d * 16 + int(num / (den * 2))                    # C division truncates toward zero
            else:
                d *= 16
            out[y, x + minx1] = d + md * 16
        for x in range(minx1, maxx1):                                # the matcher's own left-right check (:598-613)
            d1 = int(out[y, x])
            if d1 == (md - 1) * 16:
                continue
            lo, hi = d1 >> 4, (d1 + 15) >> 4
            xl, xh = x - lo, x - hi
            if (0 <= xl < W and d2p[xl] >= md and abs(d2p[xl] - lo) > maxdiff and
                    0 <= xh < W and d2p[xh] >= md and abs(d2p[xh] - hi) > maxdiff):
                out[y, x] = (md - 1) * 16
    return out


def naive_median3(a):
    H, W = a.shape
    p = np.pad(a, 1, mode="edge")
    st = np.stack([p[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)])
    return np.sort(st, 0)[4]


def _pair(seed, H, W, cn=1, shift=3):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H, W + 40) + ((cn,) if cn > 1 else ()), dtype=np.uint8)
    base = (base // 2 + np.roll(base, 1, 1) // 2).astype(np.uint8)
    return np.ascontiguousarray(base[:, 20:20 + W]), np.ascontiguousarray(np.roll(base, -shift, 1)[:, 20:20 + W])


@pytest.mark.parametrize("H,W,nd,bs,md,P1,P2,cap,ur,cn", [
    (13, 40, 16, 3, 0, 72, 288, 63, 0, 1),       # the sample's setting for a 1-channel pair: P1 = 8*cn*w*w ... here 24*3, 96*3
    (11, 37, 16, 5, 0, 0, 0, 0, 0, 1),           # every default (P1 2, P2 5, cap 15)
    (12, 45, 16, 1, -15, 10, 100, 31, 0, 1),     # the right matcher's range (minDisparity = -(0+16)+1)
    (10, 50, 32, 3, 2, 216, 864, 63, 15, 1),     # positive minimum disparity, uniqueness test on
    (9, 33, 16, 7, -5, 50, 51, 20, 5, 1),        # range straddling zero, window taller than half the image
    (8, 30, 16, 3, 0, 216, 864, 63, 0, 3),       # 3-channel views (the sample feeds colour images to SGBM)
    (7, 20, 32, 3, 0, 10, 100, 63, 0, 1),        # search range wider than the image: everything invalid
])
def test_oracle_equals_direct_statement(oracle, H, W, nd, bs, md, P1, P2, cap, ur, cn):
    a, b = _pair(H * W + nd, H, W, cn)
    got, raw = oracle.sgbm_compute(a, b, nd, bs, md, P1, P2, cap, ur, want_raw=True)
    exp_raw = naive_sgbm(a, b, nd, bs, md, P1, P2, cap, ur)
    assert np.array_equal(raw, exp_raw)
    assert np.array_equal(got, naive_median3(exp_raw))
    if W - max(md + nd, 0) + min(md, 0) <= 0:
        assert (raw == (md - 1) * 16).all()
    # the three paths of MODE_3WAY through the oracle's general multi-path code give the same map
    assert np.array_equal(oracle.sgbm_compute(a, b, nd, bs, md, P1, P2, cap, ur, mode=oracle.SGBM_MODE_3WAY_GENERIC), got)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("H,W,nd,bs,md,P1,P2,cap,ur,cn", [
    (12, 40, 16, 3, 0, 72, 288, 63, 0, 1),
    (9, 36, 16, 5, -15, 10, 100, 31, 10, 1),
    (7, 30, 16, 3, 3, 216, 864, 63, 0, 3),
    (15, 28, 16, 1, 0, 0, 0, 0, 0, 1),          # taller than the matchable area is wide: diagonals leave through the sides
])
def test_five_and_eight_path_modes_equal_direct_statement(oracle, mode, H, W, nd, bs, md, P1, P2, cap, ur, cn):
    """MODE_SGBM (left, up-left, up, up-right, right) and MODE_HH (all eight) against the direct recursion."""
    a, b = _pair(H * W + nd + mode, H, W, cn)
    got, raw = oracle.sgbm_compute(a, b, nd, bs, md, P1, P2, cap, ur, mode=mode, want_raw=True)
    exp_raw = naive_sgbm(a, b, nd, bs, md, P1, P2, cap, ur, mode=mode)
    assert np.array_equal(raw, exp_raw)
    assert np.array_equal(got, naive_median3(exp_raw))


def test_block_costs_helper(oracle):
    a, b = _pair(5, 12, 44)
    prm = oracle.sgbm_params(16, 5, -3, prefilter_cap=40)
    C = oracle.sgbm_block_costs(a, b, prm)
    assert C.shape == (12, 44 - 13 - 3, 16) and C.min() >= 0 and C.max() > 0


def test_reference_fixture_bar(oracle):
    """test_block_matching.cpp:157-238 on the reference's own Tsukuba data: <= 10 % with its parameters (P1 10, P2 100,
    uniqueness 1; its left-right check and speckle filter are post-filters this restatement does not have), and the
    sample's parameters (disparity_filtering.cpp:166-170) do at least as well as the block matcher's 20 % bar."""
    left, right, gt = load_tsukuba()
    for bs in (5, 7, 9, 11):
        d = oracle.sgbm_compute(left, right, 16, bs, 0, P1=10, P2=100, uniqueness_ratio=1)
        assert ref_error_level(gt, d) <= 10.0, bs
    d = oracle.sgbm_compute(left, right, 16, 3, 0, P1=24 * 9, P2=96 * 9, prefilter_cap=63)
    assert ref_error_level(gt, d) <= 10.0
    swapped = oracle.sgbm_compute(right, left, 16, 3, 0, P1=24 * 9, P2=96 * 9, prefilter_cap=63)
    assert ref_error_level(gt, swapped) > 30.0            # the bar bites: views in the wrong order fail it


def test_invalid_columns_and_range(oracle):
    a, b = _pair(9, 20, 90, shift=4)
    raw = oracle.sgbm_compute(a, b, 32, 3, 0, 72, 288, 63, want_raw=True)[1]
    assert (raw[:, :32] == -16).all() and (raw[:, 32:] >= 0).all() and (raw[:, 32:] <= 31 * 16).all()
    raw = oracle.sgbm_compute(b, a, 32, 3, -31, 72, 288, 63, want_raw=True)[1]      # createRightMatcher's range
    assert (raw[:, 90 - 31:] == -32 * 16).all() and (raw[:, 1:90 - 31] <= 0).all() and (raw[:, 0] == -32 * 16).all()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("md,disp12", [(0, 0), (0, 1), (0, 3), (-15, 1), (4, 2)])
def test_matchers_own_left_right_check(oracle, mode, md, disp12):
    """disp12MaxDiff (stereo_binary_sgbm.cpp:548-556, 598-613): cv::StereoSGBM::create's default (0 -> 1) leaves the
    check ON; the filter factory switches it off with 1000000.  An occluding step makes it fire."""
    rng = np.random.default_rng(40 + mode)
    H, W = 14, 70
    base = rng.integers(0, 256, (H, W + 40), dtype=np.uint8)
    a = np.ascontiguousarray(base[:, 20:20 + W])
    b = np.ascontiguousarray(base[:, 23:23 + W]).copy()
    b[:, 30:] = base[:, 20 + 39:20 + 39 + W - 30]                    # right half at a larger disparity: occlusions
    got, raw = oracle.sgbm_compute(a, b, 16, 3, md, 72, 288, 63, 0, mode=mode, want_raw=True, disp12_max_diff=disp12)
    exp_raw = naive_sgbm(a, b, 16, 3, md, 72, 288, 63, 0, mode=mode, disp12=disp12)
    assert np.array_equal(raw, exp_raw)
    off = oracle.sgbm_compute(a, b, 16, 3, md, 72, 288, 63, 0, mode=mode, want_raw=True)[1]
    if md == 0 and disp12 <= 1:
        assert (raw != off).any()                                    # the check invalidates something
    assert ((raw == off) | (raw == (md - 1) * 16)).all()             # ... and only ever invalidates
