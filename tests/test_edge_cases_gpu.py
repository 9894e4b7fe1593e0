"""Edge cases of the drop-in boundary on the device: an obstacle-free window, invalid durations,
run-to-run bit reproducibility, boundary states updated in place."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


def _setup(P, config="C1"):
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(config, P=max(P, 1), minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    return svsdf_amd, workload, w, ctx


def test_empty_cloud_is_not_an_error(built):
    """parallel_points_num == 0: the reference loop (BEO:785) adds nothing; the callback returns energy + rho sum(T)."""
    svsdf_amd, workload, w, ctx = _setup(0)
    N = len(w["T"])
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
    with pytest.raises(svsdf_amd.SvsdfError):           # before set_points it IS an error (SVSDF_ERR_NO_POINTS)
        ctx.eval_penalty(w["coeffs"], w["T"])
    ctx.set_points(np.zeros((0, 3)))
    assert ctx.num_points() == 0
    c, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"], cost0=1.5, gradT0=np.ones(N), gradC0=np.ones((6 * N, 3)))
    assert c == 1.5 and np.all(gT == 1.0) and np.all(gC == 1.0)      # += of nothing
    f, g = ctx.lmbm_evaluate(x)
    o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                   head_state=w["head_state"], tail_state=w["tail_state"])
    fo, go, c3 = o.cost_function(np.zeros((0, 3)), x, nthreads=1)
    assert abs(f - fo) <= 1e-12 * abs(fo)
    np.testing.assert_allclose(g, go, rtol=1e-10, atol=1e-10)
    assert ctx.last_costs()[0] == 0.0
    ptr, n = ctx.eval_penalty_partial(w["coeffs"], w["T"])      # the multi-rank form still hands out a (zero) partial
    assert ptr and n == 19 * N + 1
    s, t, q, idx = ctx.query_points(w["coeffs"], w["T"])
    assert len(s) == 0 and len(idx) == 0


def test_nonpositive_durations_are_rejected(built):
    svsdf_amd, workload, w, ctx = _setup(200)
    ctx.set_points(w["points"])
    for bad in (0.0, -2.5):
        T = w["T"].copy()
        T[3] = bad
        with pytest.raises(svsdf_amd.SvsdfError, match="positive"):
            ctx.eval_penalty(w["coeffs"], T)
    with pytest.raises(svsdf_amd.SvsdfError):
        ctx.eval_penalty(w["coeffs"], -w["T"])
    c, _, _ = ctx.eval_penalty(w["coeffs"], w["T"])    # the context is still usable
    assert np.isfinite(c)


@pytest.mark.parametrize("config,P", [("C2", 30000), ("C3", 20000)])
def test_evaluations_are_bit_reproducible(built, config, P):
    """No floating-point atomics anywhere on the path: cost and gradient repeat bit for bit, run to run, whatever
    launch plan (lane-group widths, GSIP bound mode) the context is in."""
    svsdf_amd, workload, w, ctx = _setup(P, config)
    ctx.set_points(w["points"])
    ref = ctx.eval_penalty(w["coeffs"], w["T"])
    for _ in range(4):
        c, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
        assert c == ref[0]
        np.testing.assert_array_equal(gT, ref[1])
        np.testing.assert_array_equal(gC, ref[2])
    ctx2 = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                  head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx2.set_points(w["points"])
    c, gT, gC = ctx2.eval_penalty(w["coeffs"], w["T"])
    assert c == ref[0]
    np.testing.assert_array_equal(gC, ref[2])
    st = ctx.stats()
    assert st["bound_mode_decided"] == 1
    assert st["gsip_bound_mode"] == (1 if config == "C3" else 0)       # the rule's choice for star / sdHorseshoe


def test_set_conditions_keeps_the_context(built):
    svsdf_amd, workload, w, ctx = _setup(500)
    ctx.set_points(w["points"])
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
    f0, _ = ctx.lmbm_evaluate(x)
    hs = w["head_state"].copy()
    hs[0, 0] += 0.4
    ctx.set_conditions(hs, w["tail_state"])
    f1, g1 = ctx.lmbm_evaluate(x)
    fresh = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                   head_state=hs, tail_state=w["tail_state"], device=0)
    fresh.set_points(w["points"])
    f2, g2 = fresh.lmbm_evaluate(x)
    assert f1 != f0 and f1 == f2
    np.testing.assert_array_equal(g1, g2)
    # the mirror keeps its context across setConditions (resident cloud, launch plan)
    opt = svsdf_amd.TrajOptimizer()
    opt.setParam(dict(rho=w["rho"], weight_p=w["weight_p"], safety_hor=w["safety_hor"], inputdata="shapes/star.obj", device=0))
    opt.setConditions(w["head_state"], w["tail_state"], len(w["T"]))
    opt.setPoints(w["points"])
    a, _ = opt.costFunctionLmbmParallel(x)
    c0 = opt._ctx
    opt.setConditions(hs, w["tail_state"], len(w["T"]))
    b, _ = opt.costFunctionLmbmParallel(x)
    assert opt._ctx is c0 and a == f0 and b == f1
