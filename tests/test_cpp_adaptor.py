"""The header-only C++ adaptor (include/adf_ximgproc.hpp): compiles without OpenCV, and on a GPU the
C++ test program -- the reference's own test idioms written against adf::ximgproc -- passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_adaptor")


def _compile():
    import oracle
    from addingdisparityfiltering_amd import _lib

    oracle.build()
    assert os.path.exists(_lib.LIB_PATH)
    src = os.path.join(ROOT, "tests", "cpp", "test_adaptor.cpp")
    deps = [src, os.path.join(ROOT, "include", "adf_ximgproc.hpp"), os.path.join(ROOT, "include", "adf_wls.h")]
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++17", "-O1", "-DADF_NO_OPENCV", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "oracle"), src,
           "-L", os.path.join(ROOT, "addingdisparityfiltering_amd"), "-ladf_wls",
           "-L", os.path.join(ROOT, "oracle"), "-ladf_oracle",
           "-Wl,-rpath,$ORIGIN/../../addingdisparityfiltering_amd", "-Wl,-rpath,$ORIGIN/../../oracle", "-o", EXE]
    subprocess.run(cmd, check=True)
    return EXE


def _typecheck_opencv_branch():
    """The cv::Mat / cv::StereoMatcher branch of the adaptor, type-checked (-fsyntax-only, never linked) against
    declaration stubs of the few cv:: names it touches (tests/cpp/opencv_stub/): the image has no OpenCV, and
    without this the branch a cv::Mat pipeline takes would never meet a compiler."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(cpp, "opencv_stub"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(cpp, "typecheck_opencv_branch.cpp")], check=True)


def test_opencv_branch_typechecks():
    _typecheck_opencv_branch()


def test_adaptor_compiles_without_opencv():
    assert os.path.exists(_compile())


@pytest.mark.gpu
def test_adaptor_program_passes_on_gpu():
    exe = _compile()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout
