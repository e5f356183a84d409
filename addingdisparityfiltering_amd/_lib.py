"""ctypes binding of libadf_wls.so (the C-ABI of include/adf_wls.h).

The product path has no CPU fallback: if the HIP library is missing or fails to
load, importing the filter API raises.  Nothing here touches oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADF_WLS_LIB selects another build of the same library (A/B experiments with compile-time knobs)
LIB_PATH = os.environ.get("ADF_WLS_LIB") or os.path.join(_HERE, "libadf_wls.so")

ADF_OK, ADF_EBADARG, ADF_ESIZE, ADF_EHIP, ADF_ENOMEM, ADF_ENODEV = range(6)
SOLVER_EXACT, SOLVER_WAVE = 0, 1
PATH_CONF_BAND, PATH_FUSED_FIRST_PASS, PATH_MERGED_PREP, PATH_SCALED_FUSED, PATH_SCALED_HALF = 1, 2, 4, 8, 16    # adf_wls_get_last_path bits (include/adf_wls.h)
DEPTH_8U, DEPTH_16S, DEPTH_32F = 0, 3, 5


class AdfError(RuntimeError):
    """Counterpart of cv::Exception raised by CV_Assert / CV_Error on the reference path."""

    def __init__(self, code, msg):
        super().__init__("adf error %d: %s" % (code, msg))
        self.code = code


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int), ("total_ms", C.c_double),
                ("alg_bytes", C.c_double), ("moved_bytes", C.c_double)]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


# every symbol include/adf_wls.h declares: (name, restype, argtypes)
_vp, _i, _d, _pd, _sz = C.c_void_p, C.c_int, C.c_double, C.c_ssize_t, C.c_size_t
_FILTER_DEV = [_vp, _i, _vp, _pd, _pd, _vp, _pd, _pd, _i, _i, _i, _vp, _pd, _pd, _vp, _pd, _pd,
               C.POINTER(Rect), _vp]
_FILTER_SCALED_DEV = [_vp, _i, _vp, _pd, _pd, _i, _i, _vp, _pd, _pd, _i, _i, _i, _vp, _pd, _pd, _vp, _pd, _pd,
                      C.POINTER(Rect), _vp]
SYMBOLS = [
    ("adf_version", _i, []),
    ("adf_last_error", C.c_char_p, []),
    ("adf_device_count", _i, []),
    ("adf_device_pci_bus_id", _i, [_i, C.c_char_p, _i]),
    ("adf_wls_create", _i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i]),
    ("adf_wls_destroy", None, [_vp]),
    ("adf_wls_set_lambda", _i, [_vp, _d]),
    ("adf_wls_get_lambda", _i, [_vp, C.POINTER(_d)]),
    ("adf_wls_set_sigma_color", _i, [_vp, _d]),
    ("adf_wls_get_sigma_color", _i, [_vp, C.POINTER(_d)]),
    ("adf_wls_set_lrc_thresh", _i, [_vp, _i]),
    ("adf_wls_get_lrc_thresh", _i, [_vp, C.POINTER(_i)]),
    ("adf_wls_set_depth_discontinuity_radius", _i, [_vp, _i]),
    ("adf_wls_get_depth_discontinuity_radius", _i, [_vp, C.POINTER(_i)]),
    ("adf_wls_set_fgs_params", _i, [_vp, _d, _i]),
    ("adf_wls_set_solver", _i, [_vp, _i]),
    ("adf_wls_get_solver", _i, [_vp, C.POINTER(_i)]),
    ("adf_wls_get_last_solver", _i, [_vp, C.POINTER(_i)]),
    ("adf_wls_get_last_path", _i, [_vp, C.POINTER(_i)]),
    ("adf_wls_filter_device", _i, _FILTER_DEV),
    ("adf_wls_filter_host", _i, _FILTER_DEV[:-1]),
    ("adf_wls_filter_scaled_device", _i, _FILTER_SCALED_DEV),
    ("adf_wls_filter_scaled_host", _i, _FILTER_SCALED_DEV[:-1]),
    ("adf_wls_get_confidence_device", _i, [_vp, _i, _vp, _pd, _vp]),
    ("adf_wls_get_confidence_host", _i, [_vp, _i, _vp, _pd]),
    ("adf_wls_get_device", _i, [_vp, C.POINTER(_i)]),
    ("adf_wls_get_roi", _i, [_vp, C.POINTER(Rect)]),
    ("adf_wls_sync", _i, [_vp, _vp]),
    ("adf_wls_workspace_bytes", _sz, [_vp]),
    ("adf_wls_profile_enable", _i, [_vp, _i]),
    ("adf_wls_profile_read", _i, [_vp, _vp, _i, C.POINTER(_i)]),
    ("adf_fgs_create", _i, [C.POINTER(_vp), _vp, _pd, _i, _i, _i, _d, _d, _d, _i, _i]),
    ("adf_fgs_create_device", _i, [C.POINTER(_vp), _vp, _pd, _i, _i, _i, _d, _d, _d, _i, _i, _vp]),
    ("adf_fgs_destroy", None, [_vp]),
    ("adf_release_cached_memory", None, []),
    ("adf_weight_table_host", _i, [C.c_float, _vp, _i]),
    ("adf_fgs_get_device", _i, [_vp, C.POINTER(_i)]),
    ("adf_fgs_filter_host", _i, [_vp, _vp, _pd, _vp, _pd, _i, _i]),
    ("adf_fgs_filter_device", _i, [_vp, _vp, _pd, _vp, _pd, _i, _i, _vp]),
    ("adf_compute_mse_host", _i, [_vp, _pd, _vp, _pd, _i, _i, C.POINTER(Rect), C.POINTER(_d)]),
    ("adf_compute_mse_device", _i, [_vp, _pd, _vp, _pd, _i, _i, C.POINTER(Rect), C.POINTER(_d), _vp]),
    ("adf_compute_bad_pixel_percent_host", _i, [_vp, _pd, _vp, _pd, _i, _i, C.POINTER(Rect), _i, C.POINTER(_d)]),
    ("adf_compute_bad_pixel_percent_device", _i, [_vp, _pd, _vp, _pd, _i, _i, C.POINTER(Rect), _i, C.POINTER(_d), _vp]),
    ("adf_get_disparity_vis_host", _i, [_vp, _pd, _vp, _pd, _i, _i, _d]),
    ("adf_get_disparity_vis_device", _i, [_vp, _pd, _vp, _pd, _i, _i, _d, _vp]),
    ("adf_bm_create", _i, [C.POINTER(_vp), _i, _i]),
    ("adf_bm_destroy", None, [_vp]),
    ("adf_bm_set_params", _i, [_vp, _i, _i, _i, _i, _i, _i]),
    ("adf_bm_get_device", _i, [_vp, C.POINTER(_i)]),
    ("adf_bm_get_params", _i, [_vp] + [C.POINTER(_i)] * 6),
    ("adf_bm_compute_device", _i, [_vp, _i, _vp, _pd, _pd, _vp, _pd, _pd, _i, _i, _vp, _pd, _pd, _vp]),
    ("adf_bm_compute_both_device", _i, [_vp, _i, _vp, _pd, _pd, _vp, _pd, _pd, _i, _i, _vp, _pd, _pd, _vp, _pd, _pd, _vp]),
    ("adf_bm_compute_host", _i, [_vp, _i, _vp, _pd, _pd, _vp, _pd, _pd, _i, _i, _vp, _pd, _pd]),
    ("adf_sgbm_create", _i, [C.POINTER(_vp), _i, _i, _i]),
    ("adf_sgbm_destroy", None, [_vp]),
    ("adf_sgbm_get_device", _i, [_vp, C.POINTER(_i)]),
    ("adf_sgbm_set_params", _i, [_vp] + [_i] * 8),
    ("adf_sgbm_get_params", _i, [_vp] + [C.POINTER(_i)] * 8),
    ("adf_sgbm_set_disp12_max_diff", _i, [_vp, _i]),
    ("adf_sgbm_get_disp12_max_diff", _i, [_vp, C.POINTER(_i)]),
    ("adf_sgbm_compute_device", _i, [_vp, _i, _vp, _pd, _pd, _vp, _pd, _pd, _i, _i, _i, _vp, _pd, _pd, _vp]),
    ("adf_sgbm_compute_host", _i, [_vp, _i, _vp, _pd, _pd, _vp, _pd, _pd, _i, _i, _i, _vp, _pd, _pd]),
]

_lib = None


def lib():
    """Load the HIP library; raise loudly if it is not built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("ADF_WLS_LIB"):     # an A/B build of older sources may lack the newest entry points
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != ADF_OK:
        raise AdfError(rc, lib().adf_last_error().decode("utf-8", "replace"))


def device_pci_bus_id(device):
    """PCI bus id string of HIP device `device` (adf_device_pci_bus_id)."""
    buf = C.create_string_buffer(64)
    check(lib().adf_device_pci_bus_id(int(device), buf, 64))
    return buf.value.decode()
