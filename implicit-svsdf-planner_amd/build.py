"""Build the gfx950 shared library (C ABI of include/svsdf_c.h) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU
box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "svsdf_api.hip")
import glob
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + [os.path.join(HERE, "..", "include", "svsdf_c.h")]
OUT = os.path.join(HERE, "libsvsdf_hip.so")

# -ffp-contract=off: the parity build rounds every operation like the reference's x86-64 build
# (no FMA contraction); see DESIGN.md "Floating-point policy".
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result", "-pthread", "-ldl"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + FLAGS + [SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
