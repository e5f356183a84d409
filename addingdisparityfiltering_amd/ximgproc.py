"""Host-side mirror of the reference's operator interface for the disparity-filter path.

Same names, argument meaning and error behaviour as cv::ximgproc in
modules/ximgproc/include/opencv2/ximgproc/disparity_filter.hpp (DF.hpp) and
edge_filter.hpp (EF.hpp), over the C-ABI in include/adf_wls.h:

    createDisparityWLSFilter(matcher_left)          DF.hpp:131  / DF.cpp:386-414
    createRightMatcher(matcher_left)                DF.hpp:139  / DF.cpp:417-449
    createDisparityWLSFilterGeneric(use_confidence) DF.hpp:149  / DF.cpp:452-455
    DisparityWLSFilter.filter(...)                  DF.hpp:75   / DF.cpp:219-298
    DisparityWLSFilter.get*/set*                    DF.hpp:90-122
    createFastGlobalSmootherFilter(...)             EF.hpp:393
    fastGlobalSmootherFilter(...)                   EF.hpp:413

Images are numpy arrays (host path: copied to the GPU and back) or torch CUDA
tensors (device path: zero-copy, asynchronous on torch's current stream).  A
leading batch dimension filters N independent, equally sized pairs in one call.
All compute runs in the HIP library; there is no CPU fallback.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import AdfError, PATH_CONF_BAND, PATH_FUSED_FIRST_PASS, PATH_MERGED_PREP, PATH_SCALED_FUSED, PATH_SCALED_HALF, Rect, SOLVER_EXACT, SOLVER_WAVE  # noqa: F401  (re-exported)

try:  # torch is optional plumbing: device memory and streams only
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _as_rect(roi):
    if roi is None:
        return None
    if isinstance(roi, Rect):
        return roi
    x, y, w, h = roi
    return Rect(int(x), int(y), int(w), int(h))


_ITEMSIZE = {np.int16: 2, np.uint8: 1, np.float32: 4}
_TORCH_DTYPE = {np.int16: torch.int16, np.uint8: torch.uint8, np.float32: torch.float32} if torch is not None else {}
_raw_stream = getattr(getattr(torch, "_C", None), "_cuda_getCurrentRawStream", None) if torch is not None else None


class _Image:
    """Pointer + strides of a (N,)H,W(,C) image held by numpy or torch."""

    def __init__(self, arr, dtype_np, what, batched, allow_channels=(1,)):
        self.keep = arr
        if _is_torch(arr):
            if not arr.is_cuda:
                arr = arr.cpu().numpy()
            else:
                if arr.dtype != _TORCH_DTYPE[dtype_np]:
                    raise AdfError(_lib.ADF_EBADARG, "%s must have dtype %s" % (what, _TORCH_DTYPE[dtype_np]))
                self.device = True
                isz = _ITEMSIZE[dtype_np]
                self.ptr = arr.data_ptr()
                self._finish(arr.shape, [q * isz for q in arr.stride()], isz, what, batched, allow_channels)
                return
        a = np.asarray(arr)
        if a.dtype != np.dtype(dtype_np):
            raise AdfError(_lib.ADF_EBADARG, "%s must have dtype %s (got %s)" % (what, np.dtype(dtype_np), a.dtype))
        self.device = False
        self.keep = a
        self.ptr = a.ctypes.data
        self._finish(a.shape, a.strides, a.itemsize, what, batched, allow_channels)

    def _finish(self, shape, strides, itemsize, what, batched, allow_channels):
        # (this runs four times per filter call: no list surgery)
        nd = len(shape)
        k = 1 if batched else 0
        if nd - k == 2:
            self.c, sc, sx = 1, itemsize, strides[k + 1]
        elif nd - k == 3:
            self.c, sc, sx = shape[k + 2], strides[k + 2], strides[k + 1]
        else:
            raise AdfError(_lib.ADF_EBADARG, "%s has an unsupported shape %s" % (what, tuple(shape)))
        self.n, self.pair_stride = (shape[0], strides[0]) if batched else (1, 0)
        self.h, self.w, self.stride = shape[k], shape[k + 1], strides[k]
        if self.n < 1 or self.h < 1 or self.w < 1:
            raise AdfError(_lib.ADF_EBADARG, "%s is empty" % what)
        if self.c not in allow_channels:
            raise AdfError(_lib.ADF_EBADARG, "%s must have %s channel(s)" % (what, " or ".join(map(str, allow_channels))))
        if sc != itemsize or sx != itemsize * self.c:
            raise AdfError(_lib.ADF_ESIZE, "%s rows must be dense (channel-interleaved, unit pixel stride)" % what)


def _out_like(img, batched, dtype_np):
    shape = (img.n, img.h, img.w) if batched else (img.h, img.w)
    if img.device:
        tdt = {np.int16: torch.int16, np.float32: torch.float32}[dtype_np]
        return torch.empty(shape, dtype=tdt, device=img.keep.device)
    return np.empty(shape, dtype_np)


def _stream_of(img):
    if img.device:
        dev = img.keep.device
        if _raw_stream is not None and dev.index is not None:      # the raw handle of torch's current stream, without a Stream object
            return _raw_stream(dev.index)
        return torch.cuda.current_stream(dev).cuda_stream
    return None


def _handle_device(getter, handle):
    dev = C.c_int(-1)
    _lib.check(getter(handle, C.byref(dev)))
    return dev.value


def _check_device(getter, handle, imgs, what, dev=None):
    """A handle's workspace lives on the device that was current when it was created (include/adf_wls.h):
    tensors of another GPU would pair it with foreign pointers and a foreign stream."""
    if dev is None:
        dev = _handle_device(getter, handle)
    for im in imgs:
        if im is not None and im.device and im.keep.device.index != dev:
            raise AdfError(_lib.ADF_EBADARG, "%s lives on cuda:%d; tensors on cuda:%s cannot be passed to it "
                                             "(create one handle per GPU)" % (what, dev, im.keep.device.index))


class DisparityFilter:
    """Main interface for all disparity map filters (DF.hpp:52-76)."""

    def filter(self, disparity_map_left, left_view, filtered_disparity_map=None,
               disparity_map_right=None, ROI=None, right_view=None):
        raise NotImplementedError


class DisparityWLSFilter(DisparityFilter):
    """Disparity map filter based on the Weighted Least Squares filter (DF.hpp:82-122)."""

    def __init__(self, use_confidence, left_offset=0, right_offset=0, top_offset=0, bottom_offset=0,
                 min_disp=0):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().adf_wls_create(C.byref(self._h), int(bool(use_confidence)), left_offset,
                                             right_offset, top_offset, bottom_offset, min_disp))
        self._use_confidence = bool(use_confidence)
        self._last = None  # (batched, device, example image) of the last filter call
        self._dev = _handle_device(_lib.lib().adf_wls_get_device, self._h)   # fixed at creation

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().adf_wls_destroy(h)
            except Exception:
                pass

    # ---- parameters (DF.hpp:90-122) ----
    def _getd(self, fn):
        v = C.c_double()
        _lib.check(fn(self._h, C.byref(v)))
        return v.value

    def _geti(self, fn):
        v = C.c_int()
        _lib.check(fn(self._h, C.byref(v)))
        return v.value

    def getLambda(self):
        return self._getd(_lib.lib().adf_wls_get_lambda)

    def setLambda(self, _lambda):
        _lib.check(_lib.lib().adf_wls_set_lambda(self._h, float(_lambda)))

    def getSigmaColor(self):
        return self._getd(_lib.lib().adf_wls_get_sigma_color)

    def setSigmaColor(self, _sigma_color):
        _lib.check(_lib.lib().adf_wls_set_sigma_color(self._h, float(_sigma_color)))

    def getLRCthresh(self):
        return self._geti(_lib.lib().adf_wls_get_lrc_thresh)

    def setLRCthresh(self, _LRC_thresh):
        _lib.check(_lib.lib().adf_wls_set_lrc_thresh(self._h, int(_LRC_thresh)))

    def getDepthDiscontinuityRadius(self):
        return self._geti(_lib.lib().adf_wls_get_depth_discontinuity_radius)

    def setDepthDiscontinuityRadius(self, _disc_radius):
        _lib.check(_lib.lib().adf_wls_set_depth_discontinuity_radius(self._h, int(_disc_radius)))

    # extensions of this implementation (no counterpart in DF.hpp)
    def setFGSParams(self, lambda_attenuation=0.25, num_iter=3):
        _lib.check(_lib.lib().adf_wls_set_fgs_params(self._h, float(lambda_attenuation), int(num_iter)))

    def setSolver(self, solver):
        _lib.check(_lib.lib().adf_wls_set_solver(self._h, int(solver)))

    def getSolver(self):
        return self._geti(_lib.lib().adf_wls_get_solver)

    def getLastSolver(self):
        return self._geti(_lib.lib().adf_wls_get_last_solver)

    def getLastPath(self):
        """PATH_* bits of the last filter call: which kernels its confidence stage took (introspection only)."""
        return self._geti(_lib.lib().adf_wls_get_last_path)

    def enableProfiling(self, on=True):
        """Bracket every kernel launch with HIP events on the caller's stream (measurement hook)."""
        _lib.check(_lib.lib().adf_wls_profile_enable(self._h, int(bool(on))))

    def readProfile(self):
        """{kernel class: dict(launches, total_ms, alg_bytes, moved_bytes)} since enableProfiling()."""
        buf = (_lib.KernelTime * 16)()
        n = C.c_int()
        _lib.check(_lib.lib().adf_wls_profile_read(self._h, buf, 16, C.byref(n)))
        return {buf[k].name.decode(): dict(launches=buf[k].launches, total_ms=buf[k].total_ms,
                                           alg_bytes=buf[k].alg_bytes, moved_bytes=buf[k].moved_bytes)
                for k in range(n.value)}

    def workspaceBytes(self):
        return int(_lib.lib().adf_wls_workspace_bytes(self._h))

    # ---- DisparityFilter::filter (DF.hpp:75) ----
    def filter(self, disparity_map_left, left_view, filtered_disparity_map=None,
               disparity_map_right=None, ROI=None, right_view=None):
        if disparity_map_left is None:
            raise AdfError(_lib.ADF_EBADARG, "disparity_map_left is empty")
        if left_view is None:
            raise AdfError(_lib.ADF_EBADARG, "left_view is empty")
        batched = len(disparity_map_left.shape) == 3
        dl = _Image(disparity_map_left, np.int16, "disparity_map_left", batched)
        gv = _Image(left_view, np.uint8, "left_view", batched, allow_channels=(1, 3))
        if gv.n != dl.n:
            raise AdfError(_lib.ADF_ESIZE, "batch sizes of disparity maps and views differ")
        # a disparity map of another resolution is resized to the view (DF.cpp:224-227, 239-247, 268-277)
        dr = None
        if disparity_map_right is not None and getattr(disparity_map_right, "size", 1) != 0:
            dr = _Image(disparity_map_right, np.int16, "disparity_map_right", batched)
            if (dr.n, dr.h, dr.w) != (dl.n, dl.h, dl.w):
                raise AdfError(_lib.ADF_ESIZE, "left and right disparity maps differ in size")  # DF.cpp:263-264
        elif self._use_confidence:
            raise AdfError(_lib.ADF_EBADARG, "disparity_map_right is required with use_confidence")  # DF.cpp:262
        if gv.device != dl.device or (dr is not None and dr.device != dl.device):
            raise AdfError(_lib.ADF_EBADARG, "inputs must all be numpy arrays or all be CUDA tensors")
        if filtered_disparity_map is None:
            filtered_disparity_map = _out_like(gv, batched, np.int16)
        out = _Image(filtered_disparity_map, np.int16, "filtered_disparity_map", batched)
        if (out.n, out.h, out.w) != (gv.n, gv.h, gv.w) or out.device != dl.device:               # DF.cpp:252,282
            raise AdfError(_lib.ADF_ESIZE, "filtered_disparity_map has the wrong size or placement")
        if dl.device:
            _check_device(None, self._h, (dl, gv, dr, out), "this DisparityWLSFilter", self._dev)
        roi = _as_rect(ROI)
        # (pointers go as plain integers: the prototypes in _lib.py say void*)
        args = (self._h, dl.n,
                dl.ptr, dl.stride, dl.pair_stride, dl.w, dl.h,
                gv.ptr, gv.stride, gv.pair_stride, gv.c, gv.w, gv.h,
                out.ptr, out.stride, out.pair_stride,
                dr.ptr if dr else None, dr.stride if dr else 0, dr.pair_stride if dr else 0,
                C.byref(roi) if roi is not None else None)
        if dl.device:
            _lib.check(_lib.lib().adf_wls_filter_scaled_device(*args, _stream_of(dl)))
        else:
            _lib.check(_lib.lib().adf_wls_filter_scaled_host(*args))
        self._last = (batched, dl.device, gv)
        return filtered_disparity_map

    def getConfidenceMap(self, pair=None):
        """CV_32F confidence map(s) of the last filter call (DF.hpp:117, DF.cpp:138)."""
        if self._last is None or not self._use_confidence:
            return np.zeros((0, 0), np.float32)  # the reference returns an empty Mat
        batched, device, ex = self._last
        pairs = range(ex.n) if pair is None else [pair]
        outs = []
        for k in pairs:
            if device:
                o = torch.empty((ex.h, ex.w), dtype=torch.float32, device=ex.keep.device)
                _lib.check(_lib.lib().adf_wls_get_confidence_device(self._h, k, C.c_void_p(o.data_ptr()),
                                                                    ex.w * 4, _stream_of(ex)))
            else:
                o = np.empty((ex.h, ex.w), np.float32)
                _lib.check(_lib.lib().adf_wls_get_confidence_host(self._h, k, C.c_void_p(o.ctypes.data), ex.w * 4))
            outs.append(o)
        if pair is not None or not batched:
            return outs[0]
        return torch.stack(outs) if device else np.stack(outs)

    def getROI(self):
        r = Rect()
        _lib.check(_lib.lib().adf_wls_get_roi(self._h, C.byref(r)))
        return (r.x, r.y, r.width, r.height)

    def sync(self, stream=None):
        _lib.check(_lib.lib().adf_wls_sync(self._h, stream))


# ---------------------------------------------------------------------------------------------
# Matchers.  cv::StereoBM / cv::StereoSGBM live in OpenCV's calib3d, which is outside the reference tree;
# the factories below only need the parameter accessors.  StereoBM additionally carries compute(), the
# published block-matching algorithm on the device (csrc/bm_matcher.hip; parity unpinned
# at the calib3d boundary, bit-exact against oracle/adf_oracle_bm.c); StereoSGBM.compute() is the semi-global matcher in
# the sample's mode (csrc/sgbm_matcher.hip, bit-exact against oracle/adf_oracle_sgbm.c).
# ---------------------------------------------------------------------------------------------
class StereoMatcher:
    def __init__(self, minDisparity=0, numDisparities=16, blockSize=3):
        self.minDisparity, self.numDisparities, self.blockSize = minDisparity, numDisparities, blockSize
        self.disp12MaxDiff, self.speckleWindowSize, self.uniquenessRatio = -1, 0, 10

    def getMinDisparity(self): return self.minDisparity
    def setMinDisparity(self, v): self.minDisparity = v
    def getNumDisparities(self): return self.numDisparities
    def setNumDisparities(self, v): self.numDisparities = v
    def getBlockSize(self): return self.blockSize
    def setBlockSize(self, v): self.blockSize = v
    def getDisp12MaxDiff(self): return self.disp12MaxDiff
    def setDisp12MaxDiff(self, v): self.disp12MaxDiff = v
    def getSpeckleWindowSize(self): return self.speckleWindowSize
    def setSpeckleWindowSize(self, v): self.speckleWindowSize = v
    def getUniquenessRatio(self): return self.uniquenessRatio
    def setUniquenessRatio(self, v): self.uniquenessRatio = v


class StereoBM(StereoMatcher):
    def __init__(self, numDisparities=0, blockSize=21):
        super().__init__(0, numDisparities if numDisparities > 0 else 64, blockSize)   # cv::StereoBM: 0 -> 64
        self.textureThreshold, self.uniquenessRatio, self.preFilterCap = 10, 15, 31
        self._h = None

    @staticmethod
    def create(numDisparities=0, blockSize=21):
        return StereoBM(numDisparities, blockSize)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().adf_bm_destroy(h)
            except Exception:
                pass

    def getTextureThreshold(self): return self.textureThreshold
    def setTextureThreshold(self, v): self.textureThreshold = v
    def getPreFilterCap(self): return self.preFilterCap
    def setPreFilterCap(self, v): self.preFilterCap = v

    def compute(self, left, right, disparity=None):
        """StereoMatcher::compute: CV_8UC1 views (H,W) or a batch (N,H,W) -> CV_16SC1 disparity*16, rejected
        pixels (minDisparity-1)*16.  torch CUDA tensors are matched where they are, asynchronously on torch's
        current stream; numpy arrays take the host entry point.  The left-right check and the speckle filter
        of cv::StereoBM are not implemented: the filter factory switches both off (DF.cpp:389-390)."""
        if self.disp12MaxDiff >= 0 and self.disp12MaxDiff < 1000000:
            raise AdfError(_lib.ADF_EBADARG, "disp12MaxDiff (left-right check inside the matcher) is not implemented")
        if self.speckleWindowSize > 0:
            raise AdfError(_lib.ADF_EBADARG, "speckle filtering is not implemented")
        batched = len(left.shape) == 3
        L = _Image(left, np.uint8, "left", batched)
        R = _Image(right, np.uint8, "right", batched)
        if (L.n, L.h, L.w) != (R.n, R.h, R.w):
            raise AdfError(_lib.ADF_ESIZE, "All the images must have the same size")
        if L.device != R.device:
            raise AdfError(_lib.ADF_EBADARG, "left and right must live on the same side (host or device)")
        if disparity is None:
            disparity = _out_like(L, batched, np.int16)
        D = _Image(disparity, np.int16, "disparity", batched)
        if (D.n, D.h, D.w) != (L.n, L.h, L.w) or D.device != L.device:
            raise AdfError(_lib.ADF_ESIZE, "disparity must match the views")
        lib = _lib.lib()
        if self._h is None:
            h = C.c_void_p()
            _lib.check(lib.adf_bm_create(C.byref(h), int(self.numDisparities), int(self.blockSize)))
            self._h = h
        _check_device(lib.adf_bm_get_device, self._h, [L, R, D], "this StereoBM")
        _lib.check(lib.adf_bm_set_params(self._h, int(self.minDisparity), int(self.numDisparities), int(self.blockSize),
                                         int(self.preFilterCap), int(self.textureThreshold), int(self.uniquenessRatio)))
        args = [self._h, L.n, C.c_void_p(L.ptr), L.stride, L.pair_stride, C.c_void_p(R.ptr), R.stride, R.pair_stride,
                L.w, L.h, C.c_void_p(D.ptr), D.stride, D.pair_stride]
        if L.device:
            _lib.check(lib.adf_bm_compute_device(*args, _stream_of(L)))
        else:
            _lib.check(lib.adf_bm_compute_host(*args))
        return disparity

    def computeBoth(self, left, right, disparity_left=None, disparity_right=None):
        """Extension: this matcher's map AND the map of createRightMatcher(self) (DF.cpp:417-431) from one launch --
        identical to `self.compute(left, right)` and `createRightMatcher(self).compute(right, left)`, with the views
        prefiltered once and both searches in one grid (worth it for one pair per call).  Device tensors only.

        The reference's right matcher keeps cv::StereoBM's default preFilterCap of 31 (DF.cpp:421-431 copy every
        parameter BUT the cap), so one shared prefilter is only the same computation when this matcher's cap is 31
        too; with any other cap the two maps are produced by the two separate computes (same results, two launches)."""
        batched = len(left.shape) == 3
        L = _Image(left, np.uint8, "left", batched)
        R = _Image(right, np.uint8, "right", batched)
        if not (L.device and R.device):
            raise AdfError(_lib.ADF_EBADARG, "computeBoth takes device tensors; use two compute() calls on the host")
        if (L.n, L.h, L.w) != (R.n, R.h, R.w):
            raise AdfError(_lib.ADF_ESIZE, "All the images must have the same size")
        if (self.disp12MaxDiff >= 0 and self.disp12MaxDiff < 1000000) or self.speckleWindowSize > 0:
            raise AdfError(_lib.ADF_EBADARG, "the matcher's own left-right check and speckle filter are not implemented")
        if disparity_left is None:
            disparity_left = _out_like(L, batched, np.int16)
        if disparity_right is None:
            disparity_right = _out_like(L, batched, np.int16)
        DL = _Image(disparity_left, np.int16, "disparity_left", batched)
        DR = _Image(disparity_right, np.int16, "disparity_right", batched)
        for D in (DL, DR):
            if (D.n, D.h, D.w) != (L.n, L.h, L.w) or not D.device:
                raise AdfError(_lib.ADF_ESIZE, "disparity maps must match the views")
        if self.preFilterCap != 31:
            if getattr(self, "_right", None) is None:
                self._right = StereoBM(1, 5)
            rm = createRightMatcher(self)
            for k in ("minDisparity", "numDisparities", "blockSize", "textureThreshold", "uniquenessRatio",
                      "preFilterCap", "disp12MaxDiff", "speckleWindowSize"):
                setattr(self._right, k, getattr(rm, k))
            self.compute(left, right, disparity_left)
            self._right.compute(right, left, disparity_right)
            return disparity_left, disparity_right
        lib = _lib.lib()
        if self._h is None:
            h = C.c_void_p()
            _lib.check(lib.adf_bm_create(C.byref(h), int(self.numDisparities), int(self.blockSize)))
            self._h = h
        _check_device(lib.adf_bm_get_device, self._h, [L, R, DL, DR], "this StereoBM")
        _lib.check(lib.adf_bm_set_params(self._h, int(self.minDisparity), int(self.numDisparities), int(self.blockSize),
                                         int(self.preFilterCap), int(self.textureThreshold), int(self.uniquenessRatio)))
        _lib.check(lib.adf_bm_compute_both_device(
            self._h, L.n, C.c_void_p(L.ptr), L.stride, L.pair_stride, C.c_void_p(R.ptr), R.stride, R.pair_stride, L.w, L.h,
            C.c_void_p(DL.ptr), DL.stride, DL.pair_stride, C.c_void_p(DR.ptr), DR.stride, DR.pair_stride, _stream_of(L)))
        return disparity_left, disparity_right


class StereoSGBM(StereoMatcher):
    """cv::StereoSGBM's accessors plus compute() on the device (csrc/sgbm_matcher.hip): the published semi-global
    algorithm with three paths (MODE_SGBM_3WAY, the mode the reference's sample selects:
    samples/disparity_filtering.cpp:166-176), five (MODE_SGBM) or eight (MODE_HH), bit-exact against
    oracle/adf_oracle_sgbm.c; parity unpinned at calib3d."""
    MODE_SGBM, MODE_HH, MODE_SGBM_3WAY = 0, 1, 2

    def __init__(self, minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, mode=0, preFilterCap=0):
        super().__init__(minDisparity, numDisparities, blockSize)
        self.P1, self.P2, self.mode, self.preFilterCap = P1, P2, mode, preFilterCap
        self.disp12MaxDiff, self.uniquenessRatio = 0, 0     # cv::StereoSGBM::create's defaults (the filter factory raises disp12MaxDiff to 1000000)
        self._h = None

    @staticmethod
    def create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
               uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=0):
        m = StereoSGBM(minDisparity, numDisparities, blockSize, P1, P2, mode, preFilterCap)
        m.disp12MaxDiff, m.uniquenessRatio, m.speckleWindowSize = disp12MaxDiff, uniquenessRatio, speckleWindowSize
        return m

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().adf_sgbm_destroy(h)
            except Exception:
                pass

    def getP1(self): return self.P1
    def setP1(self, v): self.P1 = v
    def getP2(self): return self.P2
    def setP2(self, v): self.P2 = v
    def getMode(self): return self.mode
    def setMode(self, v): self.mode = v
    def getPreFilterCap(self): return self.preFilterCap
    def setPreFilterCap(self, v): self.preFilterCap = v

    def compute(self, left, right, disparity=None):
        """StereoMatcher::compute: CV_8UC1 / CV_8UC3 views (H,W[,3]) or a batch (N,H,W[,3]) -> CV_16SC1 disparity*16,
        invalid pixels (minDisparity-1)*16.  torch CUDA tensors are matched where they are, asynchronously on torch's
        current stream; numpy arrays take the host entry point.  MODE_SGBM_3WAY (3 paths, the sample's), MODE_SGBM (5)
        and MODE_HH (8), the matcher's own left-right check (disp12MaxDiff; create's default 0 reads as 1, the filter
        factory switches it off with 1000000, DF.cpp:389); the speckle filter is not built (DF.cpp:390 sets it to 0)."""
        if self.mode not in (StereoSGBM.MODE_SGBM, StereoSGBM.MODE_HH, StereoSGBM.MODE_SGBM_3WAY):
            raise AdfError(_lib.ADF_EBADARG, "mode must be StereoSGBM.MODE_SGBM, MODE_HH or MODE_SGBM_3WAY")
        if self.speckleWindowSize > 0:
            raise AdfError(_lib.ADF_EBADARG, "speckle filtering is not implemented")
        nd = len(left.shape)
        color = nd in (3, 4) and left.shape[-1] == 3      # (H,W,3) / (N,H,W,3); a batch of 3-pixel-wide gray images is not a case
        batched = nd == (4 if color else 3)
        L = _Image(left, np.uint8, "left", batched, allow_channels=(1, 3))
        R = _Image(right, np.uint8, "right", batched, allow_channels=(1, 3))
        if (L.n, L.h, L.w, L.c) != (R.n, R.h, R.w, R.c):
            raise AdfError(_lib.ADF_ESIZE, "All the images must have the same size")
        if L.device != R.device:
            raise AdfError(_lib.ADF_EBADARG, "left and right must live on the same side (host or device)")
        if disparity is None:
            disparity = _out_like(L, batched, np.int16)
        D = _Image(disparity, np.int16, "disparity", batched)
        if (D.n, D.h, D.w) != (L.n, L.h, L.w) or D.device != L.device:
            raise AdfError(_lib.ADF_ESIZE, "disparity must match the views")
        lib = _lib.lib()
        if self._h is None:
            h = C.c_void_p()
            _lib.check(lib.adf_sgbm_create(C.byref(h), int(self.minDisparity), int(self.numDisparities), int(self.blockSize)))
            self._h = h
        _check_device(lib.adf_sgbm_get_device, self._h, [L, R, D], "this StereoSGBM")
        _lib.check(lib.adf_sgbm_set_params(self._h, int(self.minDisparity), int(self.numDisparities), int(self.blockSize),
                                           int(self.P1), int(self.P2), int(self.preFilterCap), int(self.uniquenessRatio),
                                           int(self.mode)))
        _lib.check(lib.adf_sgbm_set_disp12_max_diff(self._h, int(self.disp12MaxDiff)))
        args = [self._h, L.n, C.c_void_p(L.ptr), L.stride, L.pair_stride, C.c_void_p(R.ptr), R.stride, R.pair_stride,
                L.c, L.w, L.h, C.c_void_p(D.ptr), D.stride, D.pair_stride]
        if L.device:
            _lib.check(lib.adf_sgbm_compute_device(*args, _stream_of(L)))
        else:
            _lib.check(lib.adf_sgbm_compute_host(*args))
        return disparity


def createDisparityWLSFilter(matcher_left):
    """DF.hpp:131, DF.cpp:386-414: set the filter up from the matcher (and mutate the matcher)."""
    matcher_left.setDisp12MaxDiff(1000000)
    matcher_left.setSpeckleWindowSize(0)
    min_disp = matcher_left.getMinDisparity()
    num_disp = matcher_left.getNumDisparities()
    wsize = matcher_left.getBlockSize()
    wsize2 = wsize // 2
    if isinstance(matcher_left, StereoBM):
        matcher_left.setTextureThreshold(0)
        matcher_left.setUniquenessRatio(0)
        wls = DisparityWLSFilter(True, max(0, min_disp + num_disp) + wsize2, max(0, -min_disp) + wsize2,
                                 wsize2, wsize2, min_disp)
        wls.setDepthDiscontinuityRadius(int(math.ceil(0.33 * wsize)))
    elif isinstance(matcher_left, StereoSGBM):
        matcher_left.setUniquenessRatio(0)
        wls = DisparityWLSFilter(True, max(0, min_disp + num_disp), max(0, -min_disp), 0, 0, min_disp)
        wls.setDepthDiscontinuityRadius(int(math.ceil(0.5 * wsize)))
    else:
        raise AdfError(_lib.ADF_EBADARG, "DisparityWLSFilter natively supports only StereoBM and StereoSGBM")
    return wls


def createRightMatcher(matcher_left):
    """DF.hpp:139, DF.cpp:417-449."""
    min_disp = matcher_left.getMinDisparity()
    num_disp = matcher_left.getNumDisparities()
    wsize = matcher_left.getBlockSize()
    if isinstance(matcher_left, StereoBM):
        right = StereoBM.create(num_disp, wsize)
        right.setMinDisparity(-(min_disp + num_disp) + 1)
        right.setTextureThreshold(0)
        right.setUniquenessRatio(0)
        right.setDisp12MaxDiff(1000000)
        right.setSpeckleWindowSize(0)
        return right
    if isinstance(matcher_left, StereoSGBM):
        right = StereoSGBM.create(-(min_disp + num_disp) + 1, num_disp, wsize)
        right.setUniquenessRatio(0)
        right.setP1(matcher_left.getP1())
        right.setP2(matcher_left.getP2())
        right.setMode(matcher_left.getMode())
        right.setPreFilterCap(matcher_left.getPreFilterCap())
        right.setDisp12MaxDiff(1000000)
        right.setSpeckleWindowSize(0)
        return right
    raise AdfError(_lib.ADF_EBADARG, "createRightMatcher supports only StereoBM and StereoSGBM")


def createDisparityWLSFilterGeneric(use_confidence):
    """DF.hpp:149, DF.cpp:452-455."""
    return DisparityWLSFilter(use_confidence)


# ---------------------------------------------------------------------------------------------
# Fast Global Smoother (EF.hpp:361-413)
# ---------------------------------------------------------------------------------------------
class FastGlobalSmootherFilter:
    def __init__(self, guide, lambda_, sigma_color, lambda_attenuation=0.25, num_iter=3, solver=SOLVER_WAVE):
        self._h = C.c_void_p()
        if _is_torch(guide) and guide.is_cuda:
            # the guide already lives in HBM (a device pipeline, e.g. sparse_match_interpolators.cpp:202-203):
            # adf_fgs_create_device, asynchronous on torch's current stream, nothing crosses PCIe
            if guide.numel() == 0:
                raise AdfError(_lib.ADF_EBADARG, "guide is empty")  # FGS.cpp:143
            if guide.dtype != torch.uint8 or guide.dim() not in (2, 3) or (guide.dim() == 3 and guide.shape[2] not in (1, 3)):
                raise AdfError(_lib.ADF_EBADARG, "guide must be CV_8UC1 or CV_8UC3")  # FGS.cpp:144
            gt = guide.contiguous()
            ch = 1 if gt.dim() == 2 else gt.shape[2]
            self._shape = tuple(gt.shape[:2])
            with torch.cuda.device(gt.device):
                st = C.c_void_p(torch.cuda.current_stream(gt.device).cuda_stream)
                _lib.check(_lib.lib().adf_fgs_create_device(C.byref(self._h), C.c_void_p(gt.data_ptr()), gt.shape[1] * ch,
                                                            ch, gt.shape[1], gt.shape[0], float(lambda_), float(sigma_color),
                                                            float(lambda_attenuation), int(num_iter), int(solver), st))
            self._guide_keep = gt      # the copy into the handle is queued on the stream: keep the source alive
            return
        if guide is None or getattr(guide, "size", 0) == 0:
            raise AdfError(_lib.ADF_EBADARG, "guide is empty")  # FGS.cpp:143
        g = np.ascontiguousarray(guide)
        if g.dtype != np.uint8 or g.ndim not in (2, 3) or (g.ndim == 3 and g.shape[2] not in (1, 3)):
            raise AdfError(_lib.ADF_EBADARG, "guide must be CV_8UC1 or CV_8UC3")  # FGS.cpp:144
        ch = 1 if g.ndim == 2 else g.shape[2]
        self._shape = g.shape[:2]
        _lib.check(_lib.lib().adf_fgs_create(C.byref(self._h), C.c_void_p(g.ctypes.data), g.shape[1] * ch, ch,
                                             g.shape[1], g.shape[0], float(lambda_), float(sigma_color),
                                             float(lambda_attenuation), int(num_iter), int(solver)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().adf_fgs_destroy(h)
            except Exception:
                pass

    def filter(self, src, dst=None):
        """EF.hpp:370, FGS.cpp:182-233.  A torch CUDA tensor is filtered where it is (asynchronously on
        torch's current stream) and a CUDA tensor is returned; anything else takes the host path."""
        if _is_torch(src) and src.is_cuda:
            return self._filter_device(src, dst)
        s = np.ascontiguousarray(src)
        depth = {np.dtype(np.uint8): _lib.DEPTH_8U, np.dtype(np.int16): _lib.DEPTH_16S,
                 np.dtype(np.float32): _lib.DEPTH_32F}.get(s.dtype)
        if depth is None or s.ndim not in (2, 3):
            raise AdfError(_lib.ADF_EBADARG, "src depth must be CV_8U, CV_16S or CV_32F")  # FGS.cpp:184
        cn = 1 if s.ndim == 2 else s.shape[2]
        if cn > 4:
            raise AdfError(_lib.ADF_EBADARG, "src must have at most 4 channels")
        if s.shape[:2] != self._shape:
            raise AdfError(_lib.ADF_ESIZE,
                           "Size of the filtered image must be equal to the size of the guide image")  # FGS.cpp:187
        if dst is None:
            dst = np.empty_like(s)
        elif not (isinstance(dst, np.ndarray) and dst.flags["C_CONTIGUOUS"] and dst.shape == s.shape and dst.dtype == s.dtype):
            # the library writes h rows of w*cn elements at a dense stride: anything else would be overrun
            raise AdfError(_lib.ADF_ESIZE, "dst must be a C-contiguous ndarray with src's shape and dtype")
        rowb = s.shape[1] * cn * s.itemsize
        _lib.check(_lib.lib().adf_fgs_filter_host(self._h, C.c_void_p(s.ctypes.data), rowb,
                                                  C.c_void_p(dst.ctypes.data), rowb, depth, cn))
        return dst


    def _filter_device(self, src, dst):
        depth = {torch.uint8: _lib.DEPTH_8U, torch.int16: _lib.DEPTH_16S, torch.float32: _lib.DEPTH_32F}.get(src.dtype)
        if depth is None or src.dim() not in (2, 3):
            raise AdfError(_lib.ADF_EBADARG, "src depth must be CV_8U, CV_16S or CV_32F")  # FGS.cpp:184
        s = src.contiguous()
        cn = 1 if s.dim() == 2 else s.shape[2]
        if cn > 4:
            raise AdfError(_lib.ADF_EBADARG, "src must have at most 4 channels")
        if tuple(s.shape[:2]) != tuple(self._shape):
            raise AdfError(_lib.ADF_ESIZE,
                           "Size of the filtered image must be equal to the size of the guide image")  # FGS.cpp:187
        if dst is None:
            dst = torch.empty_like(s)
        elif not (_is_torch(dst) and dst.is_cuda and dst.is_contiguous() and dst.shape == s.shape and dst.dtype == s.dtype):
            raise AdfError(_lib.ADF_EBADARG, "dst must be a contiguous CUDA tensor shaped like src")
        rowb = s.shape[1] * cn * s.element_size()
        dev = C.c_int(-1)
        _lib.check(_lib.lib().adf_fgs_get_device(self._h, C.byref(dev)))
        if s.device.index != dev.value or dst.device.index != dev.value:
            raise AdfError(_lib.ADF_EBADARG, "this FastGlobalSmootherFilter lives on cuda:%d" % dev.value)
        st = C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream)
        _lib.check(_lib.lib().adf_fgs_filter_device(self._h, C.c_void_p(s.data_ptr()), rowb,
                                                    C.c_void_p(dst.data_ptr()), rowb, depth, cn, st))
        return dst


def createFastGlobalSmootherFilter(guide, lambda_, sigma_color, lambda_attenuation=0.25, num_iter=3, solver=SOLVER_WAVE):
    """EF.hpp:393 (`solver` is this library's extension: SOLVER_WAVE or the bit-exact SOLVER_EXACT)."""
    return FastGlobalSmootherFilter(guide, lambda_, sigma_color, lambda_attenuation, num_iter, solver)


def fastGlobalSmootherFilter(guide, src, lambda_, sigma_color, lambda_attenuation=0.25, num_iter=3, dst=None,
                             solver=SOLVER_WAVE):
    """EF.hpp:413, FGS.cpp:687-691."""
    return createFastGlobalSmootherFilter(guide, lambda_, sigma_color, lambda_attenuation, num_iter, solver).filter(src, dst)


def releaseCachedMemory():
    """Returns the device blocks of destroyed filters and the weight tables the library keeps for the next filter
    (include/adf_wls.h: adf_release_cached_memory) to the driver."""
    _lib.lib().adf_release_cached_memory()


# ---------------------------------------------------------------------------------------------
# Evaluation utilities (DF.hpp:163-204, DF.cpp:460-556)
# ---------------------------------------------------------------------------------------------
UNKNOWN_DISPARITY = 16320  # DF.cpp:460


def readGT(src_path):
    """DF.hpp:163 / DF.cpp:462-495: ground-truth disparity (x16) from a Middlebury (8-bit gray: value*16,
    0 -> unknown) or MPI-Sintel (8-bit colour: 64*R + G/4) image.  Returns (status, map); status 0 = ok,
    1 = unsupported image, like the reference.  Decoding uses Pillow (the reference uses cv::imread)."""
    try:
        from PIL import Image

        im = Image.open(src_path)
        im.load()
    except Exception:
        return 1, np.zeros((0, 0), np.int16)
    if im.mode in ("RGB", "RGBA") and im.mode == "RGB":
        a = np.asarray(im, np.int32)                       # PIL is RGB; the reference indexes BGR val[2]=R, val[1]=G
        return 0, (64 * a[:, :, 0] + a[:, :, 1] // 4).astype(np.int16)
    if im.mode == "L":
        a = np.asarray(im, np.int32)
        return 0, np.where(a == 0, UNKNOWN_DISPARITY, 16 * a).astype(np.int16)
    return 1, np.zeros((im.size[1], im.size[0]), np.int16)


def _eval_pair(GT, src, ROI):
    g = _Image(GT, np.int16, "GT", False)
    s = _Image(src, np.int16, "src", False)
    if (g.h, g.w) != (s.h, s.w):
        raise AdfError(_lib.ADF_ESIZE, "GT and src differ in size")   # DF.cpp:501
    if g.device != s.device:
        raise AdfError(_lib.ADF_EBADARG, "GT and src must both be numpy arrays or both be CUDA tensors")
    return g, s, _as_rect(ROI)


def computeMSE(GT, src, ROI=None):
    """DF.hpp:176, DF.cpp:497-517."""
    g, s, roi = _eval_pair(GT, src, ROI)
    out = C.c_double()
    args = [C.c_void_p(g.ptr), g.stride, C.c_void_p(s.ptr), s.stride, g.w, g.h,
            C.byref(roi) if roi is not None else None, C.byref(out)]
    if g.device:
        _lib.check(_lib.lib().adf_compute_mse_device(*args, _stream_of(g)))
    else:
        _lib.check(_lib.lib().adf_compute_mse_host(*args))
    return out.value


def computeBadPixelPercent(GT, src, ROI=None, thresh=24):
    """DF.hpp:190, DF.cpp:519-539."""
    g, s, roi = _eval_pair(GT, src, ROI)
    out = C.c_double()
    args = [C.c_void_p(g.ptr), g.stride, C.c_void_p(s.ptr), s.stride, g.w, g.h,
            C.byref(roi) if roi is not None else None, int(thresh), C.byref(out)]
    if g.device:
        _lib.check(_lib.lib().adf_compute_bad_pixel_percent_device(*args, _stream_of(g)))
    else:
        _lib.check(_lib.lib().adf_compute_bad_pixel_percent_host(*args))
    return out.value


def getDisparityVis(src, dst=None, scale=1.0):
    """DF.hpp:202, DF.cpp:541-556."""
    s = _Image(src, np.int16, "src", False)
    if dst is None:
        dst = torch.empty((s.h, s.w), dtype=torch.uint8, device=s.keep.device) if s.device else np.empty((s.h, s.w), np.uint8)
    d = _Image(dst, np.uint8, "dst", False)
    if (d.h, d.w) != (s.h, s.w) or d.device != s.device:
        raise AdfError(_lib.ADF_ESIZE, "dst has the wrong size or placement")
    args = [C.c_void_p(s.ptr), s.stride, C.c_void_p(d.ptr), d.stride, s.w, s.h, float(scale)]
    if s.device:
        _lib.check(_lib.lib().adf_get_disparity_vis_device(*args, _stream_of(s)))
    else:
        _lib.check(_lib.lib().adf_get_disparity_vis_host(*args))
    return dst
