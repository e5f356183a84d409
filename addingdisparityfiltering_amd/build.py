"""Build libadf_wls.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# source -> extra flags
SOURCES = {
    "adf_api.hip": [],
    "conf_kernels.hip": [],
    "weights_kernels.hip": [],
    "fgs_exact.hip": [],
    "fgs_wave_h.hip": [],
    "fgs_wave_v.hip": ["-fno-slp-vectorize"],  # see the file header
    "eval_kernels.hip": [],
    "resize_kernels.hip": [],
    "bm_matcher.hip": [],
    "sgbm_matcher.hip": [],
}
HEADERS = ["adf_internal.h", "fgs_wave_common.h"]
OUT = os.path.join(_HERE, "libadf_wls.so")
# -ffp-contract=off: the exact solver and the confidence map reproduce the reference's separate
# multiply / subtract roundings; kernels that want FMA ask for it explicitly.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def build_variant(name, defines, verbose=False):
    """Build libadf_wls_<name>.so with extra -D flags (experiments; selected with ADF_WLS_LIB)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "build", name)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(_HERE, "libadf_wls_%s.so" % name)
    procs, objs = [], []
    for src, extra in SOURCES.items():
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + extra + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True, cwd=CSRC)
    return out


def build_native(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(_HERE, "..", "include", "adf_wls.h")]
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src, extra in SOURCES.items():
        spath = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(spath), hdr_m):
            continue
        cmd = [hipcc] + FLAGS + extra + ["-c", spath, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if procs or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
