"""CPU tests of the oracle: pins the restatement with the reference's own known-answer / invariant
tests and with an independent float64 solve, before anything on the GPU is compared to it."""
import numpy as np
import pytest

from addingdisparityfiltering_amd import synthetic


def _rand_guide(rng, h, w, ch):
    shape = (h, w) if ch == 1 else (h, w, ch)
    return rng.integers(0, 255, shape, dtype=np.uint8)


def test_lut_matches_formula(oracle):
    # FGS.cpp:674 LUT[i] = -exp(-sqrt(i)/sigma) in float
    lut = oracle.lut(1.5)
    i = np.array([0, 1, 2, 100, 65025, 195075, 196607])
    ref = -np.exp(-np.sqrt(i.astype(np.float64)) / 1.5)
    np.testing.assert_allclose(lut[i], ref, rtol=3e-7, atol=1e-45)
    assert lut[0] == -1.0


def test_weights_definition(oracle):
    rng = np.random.default_rng(1)
    for ch in (1, 3):
        g = _rand_guide(rng, 23, 31, ch)
        chor, cvert = oracle.weights(g, 7.0)
        lut = oracle.lut(7.0)
        gi = g.astype(np.int64).reshape(23, 31, -1)
        ih = ((gi[:, :-1] - gi[:, 1:]) ** 2).sum(2)
        iv = ((gi[:-1] - gi[1:]) ** 2).sum(2)
        assert np.array_equal(chor[:, :-1], lut[ih]) and np.all(chor[:, -1] == 0)   # FGS.cpp:607-614
        assert np.array_equal(cvert[:-1], lut[iv]) and np.all(cvert[-1] == 0)      # FGS.cpp:635-660


def test_splat_surface_accuracy(oracle):
    """test_fgs_filter.cpp:59-87: a constant CV_16S image must come back unchanged (mean L1 <= 1/64)."""
    rnd = np.random.default_rng(0)
    for _ in range(5):
        w, h = int(rnd.integers(512, 1024)), int(rnd.integers(512, 1024))
        ch = int(rnd.choice([1, 3]))
        guide = _rand_guide(rnd, h, w, ch)
        cn = int(rnd.integers(1, 4))
        val = rnd.integers(0, 255, cn)
        src = np.broadcast_to(val.astype(np.int16), (h, w, cn)).copy()
        lam, sig = float(rnd.uniform(100, 10000)), float(rnd.uniform(1.0, 100.0))
        res = oracle.fgs_filter(guide, src, lam, sig, threads=4)
        assert np.abs(res.astype(np.int64) - src).mean() <= 1.0 / 64


@pytest.mark.parametrize("ch", [1, 3])
def test_against_float64_banded_solve(oracle, ch):
    """Independent check: LAPACK float64 solve of (I + lambda_n L_w), SURVEY 8c item 2."""
    from oracle.banded_f64 import fgs_f64

    rng = np.random.default_rng(5 + ch)
    h, w = 61, 127
    guide = _rand_guide(rng, h, w, ch)
    # smooth-ish guide so that lambda*w is large and the system is far from the identity
    guide = (guide // 32 * 32).astype(np.uint8)
    src = rng.normal(0, 1000, (h, w)).astype(np.float32)
    got = oracle.fgs_planes(guide, src[None], 8000.0, 1.5)[0]
    ref = fgs_f64(guide, src, 8000.0, 1.5)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-4, err


def test_thread_count_bit_exact_in_scalar_order(oracle):
    rng = np.random.default_rng(9)
    guide = _rand_guide(rng, 75, 90, 3)
    src = rng.normal(0, 500, (2, 75, 90)).astype(np.float32)
    a = oracle.fgs_planes(guide, src, 3000.0, 4.0, order=oracle.ORDER_SCALAR, threads=1)
    b = oracle.fgs_planes(guide, src, 3000.0, 4.0, order=oracle.ORDER_SCALAR, threads=5)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("size", [(127, 61), (320, 240)])
@pytest.mark.parametrize("ch", [1, 3])
def test_reference_simd_order_within_one_lsb(oracle, size, ch):
    """test_disparity_wls_filter.cpp:99-153: results of the reference's own evaluation orders /
    thread counts agree to NORM_INF <= 1 and L1 <= N/256.  Restated with the oracle's two orders."""
    w, h = size
    view, dl, dr, roi = synthetic.make_artificial_example(w, h, ch, seed=3)
    outs = []
    for order, threads in ((oracle.ORDER_SCALAR, 1), (oracle.ORDER_REF_SIMD, 1), (oracle.ORDER_REF_SIMD, 5)):
        p = oracle.default_params(order=order, threads=threads, sigma_color=1.5)
        outs.append(oracle.wls_filter(dl, view, dr, roi, p)[0].astype(np.int64))
    for o in outs[1:]:
        diff = np.abs(o - outs[0])
        assert diff.max() <= 1
        assert diff.sum() <= o.size / 256.0


def test_sat16_convention(oracle):
    # round-half-even + saturation; NaN / out-of-int-range -> -32768 (cvtss2si)
    assert oracle.sat16(0.5) == 0 and oracle.sat16(1.5) == 2 and oracle.sat16(2.5) == 2
    assert oracle.sat16(-0.5) == 0 and oracle.sat16(-1.5) == -2
    assert oracle.sat16(40000.0) == 32767 and oracle.sat16(-40000.0) == -32768
    assert oracle.sat16(float("nan")) == -32768
    assert oracle.sat16(float("inf")) == -32768 and oracle.sat16(3e9) == -32768
    assert oracle.sat16(2147483520.0) == 32767


def test_box_filter_reflect101(oracle):
    """DF.cpp:105-115,161-194: box mean / mean of squares on the ROI copy with BORDER_REFLECT_101."""
    rng = np.random.default_rng(4)
    H, W = 40, 53
    disp = rng.integers(-2000, 4000, (H, W)).astype(np.int16)
    for roi, r in (((5, 3, 40, 30), 2), ((0, 0, W, H), 5), ((7, 0, 9, 4), 5), ((3, 2, 1, 20), 1)):
        x, y, w, h = roi
        got = oracle.discontinuity(disp, roi, r)
        sub = disp[y:y + h, x:x + w].astype(np.int64)
        # generic reference: explicit index reflection
        def refl(p, n):
            if n == 1:
                return 0
            while p < 0 or p >= n:
                p = -p if p < 0 else 2 * (n - 1) - p
            return p
        iy = np.array([[refl(i + d, h) for d in range(-r, r + 1)] for i in range(h)])
        ix = np.array([[refl(j + d, w) for d in range(-r, r + 1)] for j in range(w)])
        win = sub[iy[:, :, None, None], ix[None, None, :, :]]          # (h, k, w, k)
        s1 = win.sum(axis=(1, 3)); s2 = (win ** 2).sum(axis=(1, 3))
        k2 = float((2 * r + 1) ** 2)
        mean = (s1 * (1.0 / k2)).astype(np.float32)
        sq = (s2 * (1.0 / k2)).astype(np.float32)
        var = sq - mean * mean
        exp = np.maximum(np.float32(1.0) - np.float32(0.001) * var, np.float32(0))
        full = np.zeros((H, W), np.float32)
        full[y:y + h, x:x + w] = exp
        assert np.array_equal(got, full)


def test_lrc_against_python_loops(oracle):
    """DF.cpp:306-341 restated as plain Python loops on a tiny image (integer logic, exhaustive)."""
    rng = np.random.default_rng(8)
    H, W = 12, 40
    roi = (6, 1, 30, 10)
    x, y, w, h = roi
    dl = rng.integers(-40, 200, (H, W)).astype(np.int16)
    dr = (-dl + rng.integers(-40, 40, (H, W))).astype(np.int16)
    r, thresh = 1, 24
    conf = oracle.confidence(dl, dr, roi, radius=r, lrc_thresh=thresh)
    cl = oracle.discontinuity(dl, roi, r)
    rrx = W - (x + w)
    cr = oracle.discontinuity(dr, (rrx, y, w, h), r)
    exp = cl.copy()
    for i in range(H):
        for j in range(x, x + w):
            ridx = j - (int(dl[i, j]) >> 4)
            if rrx <= ridx < rrx + w:
                if abs(int(dl[i, j]) + int(dr[i, ridx])) < thresh:
                    exp[i, j] = min(cl[i, j], cr[i, ridx])
                else:
                    exp[i, j] = 0.0
    exp = np.float32(255.0) * exp
    assert np.array_equal(conf, exp)
    assert np.all(conf[:, :x] == 0) and np.all(conf[:y] == 0)


def test_wls_fill_and_roi(oracle):
    view, dl, dr, _ = synthetic.make_artificial_example(96, 64, 3, seed=11)
    roi = (10, 4, 70, 50)
    out, conf = oracle.wls_filter(dl, view, dr, roi, oracle.default_params(disc_radius=2))
    mask = np.zeros_like(out, bool)
    mask[4:54, 10:80] = True
    assert np.all(out[~mask] == -16)            # DF.cpp:149,254,284
    assert np.all(conf[~mask] == 0)             # DF.cpp:187-190
    assert conf.max() <= 255.0 and conf.min() >= 0.0
    # no-confidence path: plain FGS on the int16 disparity (DF.cpp:235-259)
    out2, _ = oracle.wls_filter(dl, view, None, roi, oracle.default_params(use_confidence=0))
    assert np.all(out2[~mask] == -16)
    sub = np.ascontiguousarray(dl[4:54, 10:80])
    ref = oracle.fgs_filter(np.ascontiguousarray(view[4:54, 10:80]), sub, 8000.0, 1.0)
    assert np.array_equal(out2[4:54, 10:80], ref)


def test_zero_confidence_edge_case(oracle):
    """conf == 0 over the whole ROI: 0 * (1/(0+1e-43f)) = 0*inf = NaN -> saturate_cast gives -32768
    (the convention chosen for the reference's undefined corner)."""
    H, W = 20, 48
    view = np.full((H, W), 100, np.uint8)
    dl = np.full((H, W), 256, np.int16)   # right_idx = j - 16 lands in the right ROI [0,32) for every j
    dr = np.full((H, W), 900, np.int16)   # |dl+dr| >= thresh everywhere -> confidence 0
    roi = (16, 0, 32, 20)
    out, conf = oracle.wls_filter(dl, view, dr, roi, oracle.default_params(disc_radius=1))
    assert np.all(conf == 0)
    assert np.all(out[:, 16:] == -32768) and np.all(out[:, :16] == -16)


def test_generic_fgs_depths(oracle):
    rng = np.random.default_rng(12)
    guide = _rand_guide(rng, 33, 47, 3)
    for dt, cn in ((np.uint8, 1), (np.uint8, 3), (np.uint8, 4), (np.int16, 1), (np.int16, 3), (np.float32, 1)):
        shape = (33, 47) if cn == 1 else (33, 47, cn)
        if dt == np.float32:
            src = rng.uniform(-1e5, 1e5, shape).astype(np.float32)
        elif dt == np.int16:
            src = rng.integers(-32767, 32767, shape).astype(np.int16)
        else:
            src = rng.integers(0, 255, shape).astype(np.uint8)
        res = oracle.fgs_filter(guide, src, 900.0, 20.0)
        assert res.shape == src.shape and res.dtype == src.dtype
        # each channel is filtered independently with the same weights (FGS.cpp:200-221)
        if cn > 1:
            one = oracle.fgs_filter(guide, np.ascontiguousarray(src[..., 1]), 900.0, 20.0)
            assert np.array_equal(res[..., 1], one)


def test_refsimd_vector_code_equals_its_scalar_emulation(oracle):
    """ADF_ORDER_REF_SIMD runs four rows / four columns at a time on 128-bit vectors (what the reference's default
    build does, FGS.cpp:295-351, 516-548); every lane must perform the operations of the scalar emulation of that
    order, bit for bit, at widths / heights that leave every kind of tail."""
    rng = np.random.default_rng(11)
    for (h, w) in ((37, 53), (8, 4), (5, 3), (64, 129), (11, 2), (4, 1)):
        guide = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        src = rng.normal(0, 500, (h, w)).astype(np.float32)
        for threads in (1, 3):
            oracle.set_refsimd_rowwise(True)
            a = oracle.fgs_filter(guide, src, 8000.0, 7.0, order=oracle.ORDER_REF_SIMD, threads=threads)
            oracle.set_refsimd_rowwise(False)
            b = oracle.fgs_filter(guide, src, 8000.0, 7.0, order=oracle.ORDER_REF_SIMD, threads=threads)
            assert np.array_equal(a, b), (h, w, threads)
