"""Build libadf_wls.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["adf_api.hip", "conf_kernels.hip", "weights_kernels.hip", "fgs_exact.hip"]
OUT = os.path.join(_HERE, "libadf_wls.so")
# -ffp-contract=off: the exact solver and the confidence map reproduce the reference's separate
# multiply / subtract roundings; kernels that want FMA ask for it explicitly.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def build_native(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "adf_internal.h"), os.path.join(_HERE, "..", "include", "adf_wls.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
