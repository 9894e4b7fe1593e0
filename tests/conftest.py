import os
import sys

import pytest

# One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 / librccl under torch/lib.
# Importing torch before the library makes the library bind to that copy as well (same SONAMEs); the other order
# leaves torch with the system copy of HIP but its own HSA and fails to see the GPU.  bench.py imports torch first too.
try:
    import torch  # noqa: F401
except Exception:  # CPU-only boxes without torch still run the non-GPU tests
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "implicit-svsdf-planner_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the HIP library + oracle once per session (hipcc cross-compiles on CPU)."""
    import __graft_entry__ as ge
    ge.build()
    return True
