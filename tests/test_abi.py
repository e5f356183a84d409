"""The C-ABI library loads without a GPU and exports every symbol include/adf_wls.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "adf_wls.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(adf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from addingdisparityfiltering_amd import _lib

    names = _declared_functions()
    assert len(names) >= 25
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libadf_wls.so does not export %s" % n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(names) == bound, "ctypes table and header disagree: %s" % (set(names) ^ bound)
    assert _lib.lib().adf_version() == 100


def test_no_torch_types_in_the_boundary():
    src = open(os.path.join(ROOT, "include", "adf_wls.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)   # declarations only, not the prose
    assert "torch" not in src and "at::" not in src and "#include <hip" not in src


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "addingdisparityfiltering_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert not re.search(r'#include\s*[<"][^>"]*oracle', text), f
                assert "libadf_oracle" not in text, f


def test_fails_loudly_without_a_device():
    """No CPU fallback: on a box without a GPU the factories raise instead of computing on the host."""
    import addingdisparityfiltering_amd as adf
    from addingdisparityfiltering_amd import _lib

    if _lib.lib().adf_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(adf.AdfError) as e:
        adf.createDisparityWLSFilterGeneric(True)
    assert e.value.code == _lib.ADF_ENODEV
    with pytest.raises(adf.AdfError):
        adf.createFastGlobalSmootherFilter(np.zeros((8, 8), np.uint8), 10.0, 1.0)


def test_matcher_factories_host_logic():
    """createRightMatcher is pure host logic (DF.cpp:417-449)."""
    import addingdisparityfiltering_amd as adf

    left = adf.StereoSGBM.create(0, 160, 3)
    left.setP1(24); left.setP2(96); left.setMode(1); left.setPreFilterCap(63)
    right = adf.createRightMatcher(left)
    assert isinstance(right, adf.StereoSGBM)
    assert right.getMinDisparity() == -(0 + 160) + 1 and right.getNumDisparities() == 160
    assert (right.getP1(), right.getP2(), right.getMode(), right.getPreFilterCap()) == (24, 96, 1, 63)
    assert right.disp12MaxDiff == 1000000 and right.speckleWindowSize == 0 and right.uniquenessRatio == 0
    bm = adf.StereoBM.create(64, 15)
    rbm = adf.createRightMatcher(bm)
    assert isinstance(rbm, adf.StereoBM) and rbm.getMinDisparity() == -63 and rbm.textureThreshold == 0
    with pytest.raises(adf.AdfError):
        adf.createRightMatcher(adf.StereoMatcher())
    with pytest.raises(adf.AdfError):
        adf.createDisparityWLSFilter(adf.StereoMatcher())
    right.setMode(7)
    with pytest.raises(adf.AdfError):                       # unknown mode
        right.compute(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8))
    fresh = adf.StereoSGBM.create(0, 16, 3)
    assert (fresh.getMode(), fresh.getDisp12MaxDiff(), fresh.getUniquenessRatio()) == (0, 0, 0)   # cv::StereoSGBM::create's defaults
    fresh.setSpeckleWindowSize(50)
    with pytest.raises(adf.AdfError):                       # the speckle filter is not built
        fresh.compute(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8))


def test_synthetic_example_shape_and_determinism():
    from addingdisparityfiltering_amd import synthetic

    v, dl, dr, roi = synthetic.make_artificial_example(320, 240, 3, seed=7)
    v2, dl2, dr2, _ = synthetic.make_artificial_example(320, 240, 3, seed=7)
    assert v.shape == (240, 320, 3) and v.dtype == np.uint8 and dl.dtype == np.int16
    assert np.array_equal(v, v2) and np.array_equal(dl, dl2) and np.array_equal(dr, dr2)
    assert roi == (48, 0, 272, 240)                       # Rect(d, 0, w-d, h), P_DF:166
    assert dl.max() > 16 * 40 and dr.min() < -16 * 40     # the rectangle is there
    for cid, c in synthetic.CONFIGS.items():
        x, y, w, h = c["roi"]
        assert x + w == c["W"] and h == c["H"]


def test_weight_table_equals_the_oracles_for_a_sweep_of_sigmas():
    """The library builds the table by threads and stores the tail where the exponential has underflowed instead of
    computing it (adf_api.hip: lut_build_host): every entry must still carry the bits libm gives the oracle."""
    import ctypes as C

    import numpy as np

    import oracle
    from addingdisparityfiltering_amd import _lib

    n = 3 * 256 * 256
    sigmas = [0.05, 0.3, 0.999, 1.0, 1.5, 2.0, 3.0, 4.02, 4.03, 4.05, 4.5, 7.0, 25.0, 100.0, 1000.0]
    sigmas += list(np.exp(np.random.default_rng(0).uniform(np.log(0.2), np.log(6.0), 12)))
    for s in sigmas:
        got = np.empty(n, np.float32)
        assert _lib.lib().adf_weight_table_host(C.c_float(s), got.ctypes.data_as(C.c_void_p), n) == 0
        exp = oracle.lut(np.float32(s))
        assert np.array_equal(got.view(np.uint32), np.asarray(exp, np.float32).view(np.uint32)), s
    assert _lib.lib().adf_weight_table_host(C.c_float(1.5), None, n) != 0
    assert _lib.lib().adf_weight_table_host(C.c_float(1.5), got.ctypes.data_as(C.c_void_p), n - 1) != 0
