"""The C-ABI library loads on a CPU-only box, exports every symbol include/svsdf_c.h declares,
and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest


def test_exports_every_declared_symbol(built):
    import svsdf_amd
    from svsdf_amd import binding
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "svsdf_c.h")).read()
    declared = set(re.findall(r"\b(svsdf_[a-z_A-Z0-9]+)\s*\(", hdr))
    declared -= {"svsdf_shape_id_from_inputdata()"}
    L = ctypes.CDLL(svsdf_amd.lib_path())
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in svsdf_c.h but not exported"
    assert declared == set(binding.EXPORTS), declared ^ set(binding.EXPORTS)


def test_shape_registry_lookup(built):
    import svsdf_amd
    assert svsdf_amd.shape_id_from_inputdata("shapes/star.obj") == svsdf_amd.SHAPE_ID["star"]
    assert svsdf_amd.shape_id_from_inputdata("shapes/sdHorseshoe.obj") == svsdf_amd.SHAPE_ID["sdHorseshoe"]
    # unknown stems fall back to the Polygon (SWM:363-372)
    assert svsdf_amd.shape_id_from_inputdata("shapes/teapot.obj") == svsdf_amd.SHAPE_ID["Polygon"]


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import svsdf_amd
    with pytest.raises(svsdf_amd.SvsdfError, match="no HIP device"):
        svsdf_amd.SvsdfContext(shape="star")
    # the raw callback returns +inf and zeroes g when it cannot compute
    L = svsdf_amd.lib()
    x = np.ones(5)
    g = np.ones(5)
    f = L.svsdf_lmbm_evaluate(None, x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                              g.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 5)
    assert np.isinf(f) and not g.any()


def test_product_does_not_touch_oracle():
    """Nothing under the product package imports / links / loads the oracle."""
    root = os.path.join(os.path.dirname(__file__), "..", "implicit-svsdf-planner_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liborc" not in txt and "svsdf_oracle" not in txt and "from oracle" not in txt \
                    and "import oracle" not in txt, os.path.join(dp, f)


def test_set_conditions_on_a_host_only_context(built):
    """svsdf_set_conditions changes the boundary states in place: the host half of the callback (MINCO forward) sees
    them on the next call, without a new context."""
    import numpy as np
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C1", P=10)
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
    ctx = svsdf_amd.SvsdfContext(shape="star", head_state=w["head_state"], tail_state=w["tail_state"],
                                 flags=svsdf_amd.FLAG_HOST_ONLY)
    c0, T0 = ctx.lmbm_prepare(x)
    hs = w["head_state"].copy()
    hs[1, 0] += 0.75
    ctx.set_conditions(hs, w["tail_state"])
    c1, T1 = ctx.lmbm_prepare(x)
    ref = svsdf_amd.minco_coeffs(hs, w["tail_state"], w["q"], T1)
    np.testing.assert_array_equal(T0, T1)
    np.testing.assert_allclose(c1, ref, rtol=0, atol=1e-12)
    assert abs(c1[0, 1] - (c0[0, 1] + 0.75)) < 1e-12        # constant term of piece 0, y: the new start position
    bad = hs.copy(); bad[0, 0] = np.nan
    import pytest
    with pytest.raises(svsdf_amd.SvsdfError):
        ctx.set_conditions(bad, w["tail_state"])
