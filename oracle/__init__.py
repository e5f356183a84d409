"""ctypes front-end of the CPU oracle (oracle/adf_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libadf_oracle.so")

ORDER_SCALAR = 0
ORDER_REF_SIMD = 1
DEPTH_8U, DEPTH_16S, DEPTH_32F = 0, 3, 5
LUT_LEVELS = 3 * 256 * 256


class Params(C.Structure):
    _fields_ = [
        ("lambda_", C.c_double),
        ("sigma_color", C.c_double),
        ("use_confidence", C.c_int),
        ("lrc_thresh", C.c_int),
        ("disc_radius", C.c_int),
        ("num_iter", C.c_int),
        ("lambda_attenuation", C.c_double),
        ("order", C.c_int),
        ("threads", C.c_int),
    ]


class SGBMParams(C.Structure):
    _fields_ = [
        ("min_disparity", C.c_int),
        ("num_disparities", C.c_int),
        ("block_size", C.c_int),
        ("P1", C.c_int),
        ("P2", C.c_int),
        ("prefilter_cap", C.c_int),
        ("uniqueness_ratio", C.c_int),
        ("mode", C.c_int),
        ("disp12_max_diff", C.c_int),
    ]


SGBM_MODE_SGBM, SGBM_MODE_HH, SGBM_MODE_3WAY, SGBM_MODE_3WAY_GENERIC = 0, 1, 2, 3


class BMParams(C.Structure):
    _fields_ = [
        ("min_disparity", C.c_int),
        ("num_disparities", C.c_int),
        ("block_size", C.c_int),
        ("prefilter_cap", C.c_int),
        ("texture_threshold", C.c_int),
        ("uniqueness_ratio", C.c_int),
    ]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f))
                for f in ("adf_oracle.c", "adf_oracle_bm.c", "adf_oracle_sgbm.c", "adf_oracle.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.run(["make", "-C", _HERE, "-B", "libadf_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i, f, d, pd = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_ssize_t
        L.adf_oracle_default_params.argtypes = [C.POINTER(Params)]
        L.adf_oracle_lut.argtypes = [f, vp]
        L.adf_oracle_weights.argtypes = [vp, pd, i, i, i, vp, vp, vp, i]
        L.adf_oracle_hpass.argtypes = [vp, vp, vp, i, i, f, i, i]
        L.adf_oracle_vpass.argtypes = [vp, vp, vp, i, i, f, i, i]
        L.adf_oracle_fgs_planes.argtypes = [vp, pd, i, i, i, vp, i, d, d, d, i, i, i]
        L.adf_oracle_fgs_planes.restype = i
        L.adf_oracle_fgs_filter.argtypes = [vp, pd, i, i, i, vp, vp, i, i, d, d, d, i, i, i]
        L.adf_oracle_fgs_filter.restype = i
        L.adf_oracle_discontinuity.argtypes = [vp, pd, i, i, i, i, i, i, i, f, vp, i]
        L.adf_oracle_confidence.argtypes = [vp, pd, vp, pd, i, i, i, i, i, i, i, i, f, vp, i]
        L.adf_oracle_wls_filter.argtypes = [C.POINTER(Params), vp, pd, vp, pd, i, i, i, vp, pd,
                                            i, i, i, i, vp, pd, vp]
        L.adf_oracle_wls_filter.restype = i
        L.adf_oracle_compute_mse.argtypes = [vp, vp, i, i, i, i, i, i]
        L.adf_oracle_compute_mse.restype = d
        L.adf_oracle_bad_pixel_percent.argtypes = [vp, vp, i, i, i, i, i, i, i]
        L.adf_oracle_bad_pixel_percent.restype = d
        L.adf_oracle_disparity_vis.argtypes = [vp, vp, i, i, d]
        L.adf_oracle_resize_linear_16s.argtypes = [vp, i, i, vp, i, i, f]
        L.adf_oracle_resize_linear_32f.argtypes = [vp, i, i, vp, i, i]
        L.adf_oracle_wls_filter_scaled.argtypes = [C.POINTER(Params), vp, vp, i, i, vp, i, i, i, i, i, i, i, vp, vp]
        L.adf_oracle_wls_filter_scaled.restype = i
        L.adf_oracle_bm_prefilter_xsobel.argtypes = [vp, pd, i, i, i, vp]
        L.adf_oracle_bm_compute.argtypes = [C.POINTER(BMParams), vp, pd, vp, pd, i, i, vp, pd]
        L.adf_oracle_bm_compute.restype = i
        L.adf_oracle_sgbm_signals.argtypes = [vp, pd, i, i, i, i, vp]
        L.adf_oracle_sgbm_block_costs.argtypes = [C.POINTER(SGBMParams), vp, pd, vp, pd, i, i, i, vp]
        L.adf_oracle_sgbm_block_costs.restype = i
        L.adf_oracle_sgbm_compute.argtypes = [C.POINTER(SGBMParams), vp, pd, vp, pd, i, i, i, vp, pd, vp]
        L.adf_oracle_sgbm_compute.restype = i
        L.adf_oracle_median3_16s.argtypes = [vp, pd, vp, pd, i, i]
        L.adf_oracle_set_refsimd_rowwise.argtypes = [i]
        L.adf_oracle_sat16.argtypes = [f]
        L.adf_oracle_sat16.restype = C.c_int16
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw):
    p = Params()
    lib().adf_oracle_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, "lambda_" if k == "lambda" else k, v)
    return p


def lut(sigma):
    out = np.empty(LUT_LEVELS, np.float32)
    lib().adf_oracle_lut(float(sigma), _p(out))
    return out


def _guide(guide):
    g = np.ascontiguousarray(guide, np.uint8)
    ch = 1 if g.ndim == 2 else g.shape[2]
    h, w = g.shape[:2]
    return g, ch, w, h, w * ch


def weights(guide, sigma, threads=1):
    g, ch, w, h, stride = _guide(guide)
    chor = np.empty((h, w), np.float32)
    cvert = np.empty((h, w), np.float32)
    table = lut(sigma)
    lib().adf_oracle_weights(_p(g), stride, ch, w, h, _p(table), _p(chor), _p(cvert), threads)
    return chor, cvert


def hpass(cur, chor, lam, order=ORDER_SCALAR, threads=1):
    cur = np.array(cur, np.float32, order="C")
    h, w = cur.shape
    inter = np.empty_like(cur)
    lib().adf_oracle_hpass(_p(cur), _p(np.ascontiguousarray(chor, np.float32)), _p(inter), w, h,
                           float(lam), order, threads)
    return cur, inter


def vpass(cur, cvert, lam, order=ORDER_SCALAR, threads=1):
    cur = np.array(cur, np.float32, order="C")
    h, w = cur.shape
    inter = np.empty_like(cur)
    lib().adf_oracle_vpass(_p(cur), _p(np.ascontiguousarray(cvert, np.float32)), _p(inter), w, h,
                           float(lam), order, threads)
    return cur, inter


def fgs_planes(guide, planes, lam, sigma, atten=0.25, num_iter=3, order=ORDER_SCALAR, threads=1):
    """planes: (n, h, w) float32 filtered with one shared set of weights."""
    g, ch, w, h, stride = _guide(guide)
    pl = np.array(planes, np.float32, order="C")
    assert pl.shape[1:] == (h, w)
    rc = lib().adf_oracle_fgs_planes(_p(g), stride, ch, w, h, _p(pl), pl.shape[0], lam, sigma,
                                     atten, num_iter, order, threads)
    if rc:
        raise ValueError("adf_oracle_fgs_planes rc=%d" % rc)
    return pl


def fgs_filter(guide, src, lam, sigma, atten=0.25, num_iter=3, order=ORDER_SCALAR, threads=1):
    """fastGlobalSmootherFilter(guide, src, dst, ...) restated (EF.hpp:413)."""
    g, ch, w, h, stride = _guide(guide)
    s = np.ascontiguousarray(src)
    depth = {np.dtype(np.uint8): DEPTH_8U, np.dtype(np.int16): DEPTH_16S,
             np.dtype(np.float32): DEPTH_32F}[s.dtype]
    channels = 1 if s.ndim == 2 else s.shape[2]
    dst = np.empty_like(s)
    rc = lib().adf_oracle_fgs_filter(_p(g), stride, ch, w, h, _p(s), _p(dst), depth, channels,
                                     lam, sigma, atten, num_iter, order, threads)
    if rc:
        raise ValueError("adf_oracle_fgs_filter rc=%d" % rc)
    return dst


def discontinuity(disp, roi, radius, roll_off=0.001, threads=1):
    d = np.ascontiguousarray(disp, np.int16)
    H, W = d.shape
    out = np.empty((H, W), np.float32)
    lib().adf_oracle_discontinuity(_p(d), W * 2, W, H, roi[0], roi[1], roi[2], roi[3], radius,
                                   float(np.float32(roll_off)), _p(out), threads)
    return out


def confidence(dispL, dispR, roi, radius=5, lrc_thresh=24, resize_factor=1.0, threads=1):
    dl = np.ascontiguousarray(dispL, np.int16)
    dr = np.ascontiguousarray(dispR, np.int16)
    H, W = dl.shape
    out = np.empty((H, W), np.float32)
    lib().adf_oracle_confidence(_p(dl), W * 2, _p(dr), W * 2, W, H, roi[0], roi[1], roi[2],
                                roi[3], radius, lrc_thresh, resize_factor, _p(out), threads)
    return out


def wls_filter(dispL, guide, dispR, roi, params=None, want_conf=True):
    """DisparityWLSFilter::filter restated; returns (filtered int16, confidence float32|None)."""
    p = params if params is not None else default_params()
    dl = np.ascontiguousarray(dispL, np.int16)
    H, W = dl.shape
    g, ch, gw, gh, gstride = _guide(guide)
    assert (gw, gh) == (W, H)
    dr = None if dispR is None else np.ascontiguousarray(dispR, np.int16)
    out = np.empty((H, W), np.int16)
    conf = np.empty((H, W), np.float32) if want_conf else None
    rc = lib().adf_oracle_wls_filter(C.byref(p), _p(dl), W * 2, _p(g), gstride, ch, W, H,
                                     None if dr is None else _p(dr), W * 2,
                                     roi[0], roi[1], roi[2], roi[3], _p(out), W * 2,
                                     None if conf is None else _p(conf))
    if rc:
        raise ValueError("adf_oracle_wls_filter rc=%d" % rc)
    return out, conf


def set_refsimd_rowwise(on):
    """Test hook: run ORDER_REF_SIMD as its scalar emulation instead of the 128-bit vector code."""
    lib().adf_oracle_set_refsimd_rowwise(int(bool(on)))


def sat16(v):
    return int(lib().adf_oracle_sat16(float(v)))


def compute_mse(gt, src, roi):
    g = np.ascontiguousarray(gt, np.int16); s = np.ascontiguousarray(src, np.int16)
    H, W = g.shape
    return lib().adf_oracle_compute_mse(_p(g), _p(s), W, H, roi[0], roi[1], roi[2], roi[3])


def bad_pixel_percent(gt, src, roi, thresh=24):
    g = np.ascontiguousarray(gt, np.int16); s = np.ascontiguousarray(src, np.int16)
    H, W = g.shape
    return lib().adf_oracle_bad_pixel_percent(_p(g), _p(s), W, H, roi[0], roi[1], roi[2], roi[3], thresh)


def disparity_vis(src, scale=1.0):
    s = np.ascontiguousarray(src, np.int16)
    out = np.empty(s.shape, np.uint8)
    lib().adf_oracle_disparity_vis(_p(s), _p(out), s.shape[1], s.shape[0], float(scale))
    return out


def resize_linear(src, dsize, post_scale=1.0):
    """cv::resize(src, dsize=(w, h), INTER_LINEAR) restated for int16 / float32 single-channel images."""
    a = np.ascontiguousarray(src)
    sh, sw = a.shape
    dw, dh = dsize
    out = np.empty((dh, dw), a.dtype)
    if a.dtype == np.int16:
        lib().adf_oracle_resize_linear_16s(_p(a), sw, sh, _p(out), dw, dh, float(post_scale))
    else:
        lib().adf_oracle_resize_linear_32f(_p(np.ascontiguousarray(a, np.float32)), sw, sh, _p(out), dw, dh)
    return out


def wls_filter_scaled(dispL, guide, dispR, roi, params=None):
    """DisparityWLSFilter::filter with low-resolution disparity maps (ROI in their coordinates)."""
    p = params if params is not None else default_params()
    dl = np.ascontiguousarray(dispL, np.int16)
    dH, dW = dl.shape
    g, ch, W, H, _ = _guide(guide)
    dr = None if dispR is None else np.ascontiguousarray(dispR, np.int16)
    out = np.empty((H, W), np.int16)
    conf = np.empty((H, W), np.float32)
    rc = lib().adf_oracle_wls_filter_scaled(C.byref(p), _p(dl), None if dr is None else _p(dr), dW, dH, _p(g), ch, W, H,
                                            roi[0], roi[1], roi[2], roi[3], _p(out), _p(conf))
    if rc:
        raise ValueError("adf_oracle_wls_filter_scaled rc=%d" % rc)
    return out, conf


def bm_prefilter_xsobel(img, cap=31):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    lib().adf_oracle_bm_prefilter_xsobel(_p(img), img.strides[0], W, H, cap, _p(out))
    return out


def bm_compute(left, right, num_disparities, block_size, min_disparity=0, prefilter_cap=31,
               texture_threshold=0, uniqueness_ratio=0):
    """Block matcher restated from the published StereoBM algorithm (adf_oracle_bm.c; parity unpinned)."""
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    assert left.shape == right.shape and left.ndim == 2
    H, W = left.shape
    out = np.empty((H, W), np.int16)
    prm = BMParams(min_disparity, num_disparities, block_size, prefilter_cap, texture_threshold, uniqueness_ratio)
    rc = lib().adf_oracle_bm_compute(C.byref(prm), _p(left), left.strides[0], _p(right), right.strides[0], W, H, _p(out), W)
    if rc:
        raise ValueError("adf_oracle_bm_compute: bad arguments (%d)" % rc)
    return out


def _sgbm_images(img1, img2):
    a = np.ascontiguousarray(img1, dtype=np.uint8)
    b = np.ascontiguousarray(img2, dtype=np.uint8)
    assert a.shape == b.shape and a.ndim in (2, 3)
    cn = 1 if a.ndim == 2 else a.shape[2]
    H, W = a.shape[:2]
    return a, b, cn, W, H


def sgbm_params(num_disparities, block_size, min_disparity=0, P1=0, P2=0, prefilter_cap=0, uniqueness_ratio=0,
                mode=SGBM_MODE_3WAY, disp12_max_diff=1000000):
    return SGBMParams(min_disparity, num_disparities, block_size, P1, P2, prefilter_cap, uniqueness_ratio, mode,
                      disp12_max_diff)


def sgbm_signals(img, prefilter_cap):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    cn = 1 if a.ndim == 2 else a.shape[2]
    H, W = a.shape[:2]
    rec = np.empty((H, W, 2 * cn, 3), np.uint8)
    lib().adf_oracle_sgbm_signals(_p(a), a.strides[0], cn, W, H, prefilter_cap, _p(rec))
    return rec


def sgbm_block_costs(img1, img2, prm):
    """C[H][width1][D] of the whole image (small images only)."""
    a, b, cn, W, H = _sgbm_images(img1, img2)
    maxd = prm.min_disparity + prm.num_disparities
    w1 = (W + min(prm.min_disparity, 0)) - max(maxd, 0)
    out = np.zeros((H, max(w1, 0), prm.num_disparities), np.int16)
    rc = lib().adf_oracle_sgbm_block_costs(C.byref(prm), _p(a), a.strides[0], _p(b), b.strides[0], cn, W, H, _p(out))
    if rc:
        raise ValueError("adf_oracle_sgbm_block_costs: bad arguments (%d)" % rc)
    return out


def sgbm_compute(img1, img2, num_disparities, block_size, min_disparity=0, P1=0, P2=0, prefilter_cap=0,
                 uniqueness_ratio=0, mode=SGBM_MODE_3WAY, want_raw=False, disp12_max_diff=1000000):
    """Semi-global matcher restated from the published algorithm (adf_oracle_sgbm.c; parity unpinned)."""
    a, b, cn, W, H = _sgbm_images(img1, img2)
    prm = sgbm_params(num_disparities, block_size, min_disparity, P1, P2, prefilter_cap, uniqueness_ratio, mode,
                      disp12_max_diff)
    out = np.empty((H, W), np.int16)
    raw = np.empty((H, W), np.int16) if want_raw else None
    rc = lib().adf_oracle_sgbm_compute(C.byref(prm), _p(a), a.strides[0], _p(b), b.strides[0], cn, W, H, _p(out), W,
                                       None if raw is None else _p(raw))
    if rc:
        raise ValueError("adf_oracle_sgbm_compute: bad arguments (%d)" % rc)
    return (out, raw) if want_raw else out


def median3_16s(src):
    s = np.ascontiguousarray(src, np.int16)
    out = np.empty_like(s)
    lib().adf_oracle_median3_16s(_p(s), s.shape[1], _p(out), s.shape[1], s.shape[1], s.shape[0])
    return out
