"""The C oracle against golden vectors produced by an independently written pure-Python restatement
of the reference path (tests/golden/make_golden.py; same libm, no FMA => agreement to the last
bits is expected), and -- on the GPU -- the HIP path against the same vectors."""
import json
import os

import numpy as np
import pytest

from oracle import orc

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_small.json")))["cases"]


def _states(case):
    hs = np.array(case["head"]).T  # rows pos/vel/acc -> columns
    ts = np.array(case["tail"]).T
    return hs, ts


@pytest.mark.parametrize("case", GOLD, ids=[c["shape"] for c in GOLD])
def test_oracle_reproduces_python_restatement(case):
    hs, ts = _states(case)
    o = orc.Oracle(case["shape"], safety_hor=case["safety_hor"], weight_p=case["weight_p"], rho=case["rho"],
                   poly_params=case["poly_params"], polygon=case["polygon"], head_state=hs, tail_state=ts)
    pts = np.array(case["points"])
    x = np.array(case["x"])
    f, g, c3 = o.cost_function(pts, x)
    assert abs(f - case["cost"]) <= 1e-12 * abs(case["cost"])
    np.testing.assert_allclose(c3, case["costs3"], rtol=1e-12)
    np.testing.assert_allclose(g, case["grad"], rtol=1e-10, atol=1e-10)
    # MINCO coefficients and per-point results of getTrueSDFofSweptVolume
    coeffs = np.array(case["coeffs"])
    np.testing.assert_allclose(orc.minco_coeffs(hs, ts, np.array(case["x"][case["N"]:]).reshape(-1, 3), case["T"]),
                               coeffs, rtol=0, atol=1e-13)
    o.set_traj(coeffs, case["T"])
    sdf, tstar, grad = o.query(pts)
    per = np.array(case["per_point"])
    np.testing.assert_allclose(sdf, per[:, 0], rtol=0, atol=1e-13)
    np.testing.assert_allclose(tstar, per[:, 1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(grad, per[:, 2:4], rtol=0, atol=1e-9)
    assert o.counters()["solves"] == case["n_solves"]
    assert (per[:, 0] <= 0).sum() >= 1  # the case exercises the GSIP branch


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD, ids=[c["shape"] for c in GOLD])
def test_hip_matches_python_restatement(built, case):
    import svsdf_amd
    hs, ts = _states(case)
    ctx = svsdf_amd.SvsdfContext(shape=case["shape"], safety_hor=case["safety_hor"], weight_p=case["weight_p"],
                                 rho=case["rho"], poly_params=case["poly_params"], polygon=case["polygon"],
                                 head_state=hs, tail_state=ts, device=0)
    pts = np.array(case["points"])
    ctx.set_points(pts)
    f, g = ctx.lmbm_evaluate(np.array(case["x"]))
    assert abs(f - case["cost"]) <= 1e-7 * abs(case["cost"])
    assert np.linalg.norm(g - np.array(case["grad"])) <= 1e-5 * np.linalg.norm(case["grad"])
    sdf, tstar, grad, _ = ctx.query_points(np.array(case["coeffs"]), case["T"])
    per = np.array(case["per_point"])
    np.testing.assert_allclose(sdf, per[:, 0], rtol=0, atol=1e-8)
    np.testing.assert_allclose(tstar, per[:, 1], rtol=0, atol=1e-6)
