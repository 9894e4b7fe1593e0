"""Launch plan and fused GSIP tail (round 4): every plan returns the same bits; the plan is deterministic and observable
(svsdf_get_plan / svsdf_set_plan, include/svsdf_c.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(w, **kw):
    import svsdf_amd
    c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                               poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                               tail_state=w["tail_state"], device=0, **kw)
    c.set_points(w["points"])
    return c


def _all(c, w):
    pen = c.eval_penalty(w["coeffs"], w["T"])
    q = c.query_points(w["coeffs"], w["T"])
    return pen, q


def _same(a, b, what):
    (pa, qa), (pb, qb) = a, b
    assert pa[0] == pb[0] and np.array_equal(pa[1], pb[1]) and np.array_equal(pa[2], pb[2]), what
    for u, v in zip(qa[:3], qb[:3]):
        assert np.array_equal(u, v), what


@pytest.mark.parametrize("config,P", [("C1", 8000), ("C3", 6000), ("C5", 3000), ("C2", 30000)])
def test_fused_tail_is_invisible(built, config, P):
    """k_tail (all GSIP iterations of a batch in one launch, a half-wave owns a point) against the launch chain: at every
    starting iteration, in every GSIP bound mode, the per-point results, cost and gradients are the same bits."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(config, P=P, minco=svsdf_amd.minco_coeffs)
    ref_c = _ctx(w)
    ref_c.set_plan(tail_iter=-2)
    ref = _all(ref_c, w)
    assert ref_c.stats()["tail_iter"] == -1
    for mode in (0, 1, 2, 3):
        for ti in (0, 1, 3):
            c = _ctx(w)
            c.set_plan(bound_mode=mode, tail_iter=ti)
            for _ in range(2):      # the second evaluation sizes the tail's grid from the first one's counts
                got = _all(c, w)
            st = c.stats()
            assert st["tail_iter"] == ti and st["tail_launches"] >= 1, (mode, ti, st)
            _same(got, ref, (config, mode, ti))
            c.close()
    ref_c.close()


def test_plan_is_deterministic_and_observable(built):
    """Rules, not timings: two fresh contexts on the same cloud report the same plan, settled after two evaluations; a
    pinned plan is reported back and changes no bit; the default of a small cloud is the whole GSIP loop in the tail."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C3", P=40000, minco=svsdf_amd.minco_coeffs)
    plans = []
    for _ in range(2):
        c = _ctx(w)
        assert c.get_plan()["settled"] == 0
        n = 0
        while n < 8 and not c.get_plan()["settled"]:     # cheap bound -> full scans -> (anchor trial: two more) -> widths
            c.eval_penalty(w["coeffs"], w["T"])
            n += 1
        assert 2 <= n <= 5, n
        ref = _all(c, w)
        pl = c.get_plan()
        assert pl["settled"] == 1 and c.stats()["plan_settled"] == 1
        plans.append((pl, n))
        c.close()
    assert plans[0] == plans[1], plans
    plans = [p_[0] for p_ in plans]
    # sdHorseshoe: full scans pay, the anchor variant saves too few of them (decided by counters, not by timing)
    assert plans[0]["bound_mode"] == 1 and plans[0]["batches"] == 1 and plans[0]["lanes_per_query"] == 8
    c = _ctx(w)
    c.set_plan(bound_mode=2, batches=3, lanes_per_query=4, tail_iter=2)
    got = _all(c, w)
    pl = c.get_plan()
    assert (pl["bound_mode"], pl["batches"], pl["lanes_per_query"], pl["tail_iter"]) == (2, 3, 4, 2), pl
    assert c.stats()["batches"] == 3 and c.stats()["gsip_bound_mode"] == 2 and c.stats()["tail_iter"] == 2
    _same(got, ref, "pinned plan")
    c.set_plan()          # everything back to the rules
    for _ in range(5):
        got = _all(c, w)
    assert c.get_plan() == plans[0], (c.get_plan(), plans[0])
    _same(got, ref, "rules again")
    with pytest.raises(svsdf_amd.SvsdfError):
        c.set_plan(lanes_per_query=3)
    c.close()
    # a small cloud: the rule puts the whole GSIP loop into k_tail from the second evaluation on
    ws = workload.make("C1", P=5000, minco=svsdf_amd.minco_coeffs)
    c = _ctx(ws)
    c.eval_penalty(ws["coeffs"], ws["T"])
    assert c.stats()["tail_iter"] == -1
    c.eval_penalty(ws["coeffs"], ws["T"])
    assert c.stats()["tail_iter"] == 0 and c.stats()["tail_points"] == c.stats()["interior_points"]
    c.close()


def test_measured_batch_count_settles(built):
    """batches = -2: the count is measured (three HIP-event timings per candidate) instead of set by rule; it settles, is
    one of the candidates, and the results stay the same bits throughout."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C3", P=420000, minco=svsdf_amd.minco_coeffs)
    c = _ctx(w)
    ref = c.eval_penalty(w["coeffs"], w["T"])
    c.set_plan(batches=-2)
    n = 0
    while n < 16:
        got = c.eval_penalty(w["coeffs"], w["T"])
        assert got[0] == ref[0] and np.array_equal(got[2], ref[2])
        n += 1
        if c.get_plan()["settled"]:
            break
    assert c.get_plan()["settled"] == 1 and 10 <= n <= 15, n     # decide (+ anchor trial) + learn + 3 x 3 timed
    assert c.get_plan()["batches"] in (1, 3, 4)
    c.close()


def test_anchor_scans_are_chosen_by_counters_and_invisible(built):
    """sdHeart (BASELINE config 4): full scans pay, and the anchor variant (every third GSIP sample scanned, the others only
    if their Lipschitz bound from the anchors reaches the selection band) saves a third of the table evaluations -- the
    library keeps it, by a rule on its own counters; the result is the same bits as the full mode."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C4", P=60000, minco=svsdf_amd.minco_coeffs)
    c = _ctx(w)
    n = 0
    while n < 8 and not c.get_plan()["settled"]:
        c.eval_penalty(w["coeffs"], w["T"])
        n += 1
    assert c.get_plan()["bound_mode"] == 3 and n <= 5, (c.get_plan(), n)
    got = _all(c, w)
    anchor_evals = c.stats()["round_scan_evals"]
    f = _ctx(w)
    f.set_plan(bound_mode=1)
    ref = _all(f, w)
    assert f.stats()["gsip_bound_mode"] == 1 and anchor_evals <= 0.72 * f.stats()["round_scan_evals"]
    _same(got, ref, "anchor vs full")
    c.close(); f.close()
