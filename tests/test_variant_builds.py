"""The diagnostic / A-B build flags of the kernels keep compiling (hipcc cross-compiles gfx950 without a GPU):
-DSVSDF_SITE_STATS (in-kernel site and phase counters, tools/site_stats.py) and -DSVSDF_ELASTIC=0 (the fixed-ladder form
of the descent that rounds 1-2 measured).  One shape slice of the fast build each, device code only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "implicit-svsdf-planner_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
@pytest.mark.parametrize("flag", ["-DSVSDF_SITE_STATS", "-DSVSDF_ELASTIC=0"])
def test_kernel_build_flags_compile(tmp_path, flag):
    hipcc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    out = str(tmp_path / "slice.o")
    cmd = [hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only",
           "-DSVSDF_FAST_BUILD", "-DSVSDF_SLICE=2", flag, "-c", os.path.join(CSRC, "svsdf_shape_slice.hip"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert os.path.getsize(out) > 0
