"""Round 5 additions, on the GPU through the C ABI:

* the evaluation's tail (`k_reduce`): one block (<= 256 points, the reference's own scale), the fused <= 4-block form and
  the k_final / k_finish form give the oracle's sums at the north-star gates and repeat bit for bit;
* reference scale: the three demo maps through the producer, 24 pieces, generic durations -- full callback against the
  oracle, the whole evaluation in `k_tail` (wave-local GSIP state) identical to the launch chain and to the global-memory
  form of the tail;
* multi-device contexts: per-stripe statistics and plans (`svsdf_group_stripe`), the serial diagnostic mode, the fan-out
  time; what an 8-GPU run is judged by can be read per stripe;
* the stream pool: a context created after another one was destroyed gets the same streams back;
* the shape self-check (1-Lipschitz) and what switches off without it;
* the kernels' own clock measurement.
"""
import os

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
NT = min(os.cpu_count() or 1, 32)


def _rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


def _ctx(w, **kw):
    import svsdf_amd
    return svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                  poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                                  tail_state=w["tail_state"], device=0, **kw)


def _oracle(w):
    return orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                      poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"])


@pytest.mark.parametrize("P", [1, 37, 256, 257, 1024, 1025, 5000])
def test_every_form_of_the_reduction_matches_the_oracle(built, P):
    """k_reduce: one block / up to four blocks (last block sums) / assembly + k_final + k_finish -- the boundaries 256 | 257
    and 1024 | 1025 points included.  Sums against the oracle (cost 1e-7, gradients 1e-5), += semantics, repeatability."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C1", P=P, minco=svsdf_amd.minco_coeffs)
    c = _ctx(w)
    c.set_points(w["points"])
    o = _oracle(w)
    o.set_traj(w["coeffs"], w["T"])
    ocost, ogT, ogC = o.penalty(w["points"], nthreads=NT, sum_mode=1)
    first = None
    for k in range(4):       # (the plan moves over the first evaluations: chain -> tail; the sums must not)
        cost, gT, gC = c.eval_penalty(w["coeffs"], w["T"])
        if first is None:
            first = (cost, gT.copy(), gC.copy())
        assert cost == first[0] and np.array_equal(gT, first[1]) and np.array_equal(gC, first[2]), (P, k)
    if ocost > 0:
        assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
        assert _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5
    else:
        assert cost == 0.0 and not gC.any() and not gT.any()
    N = len(w["T"])
    c2, gT2, gC2 = c.eval_penalty(w["coeffs"], w["T"], cost0=3.5, gradT0=np.ones(N), gradC0=np.full((6 * N, 3), 2.0))
    assert c2 == 3.5 + cost and np.array_equal(gT2, 1.0 + gT) and np.array_equal(gC2, 2.0 + gC)
    c.close()


@pytest.mark.parametrize("name", ["star", "sdHorseshoe", "sdHeart"])
def test_reference_scale_callback(built, name, monkeypatch):
    """The regime the reference runs (plan_manager.cpp:156-175): demo map -> producer -> ~ 100 query points, 24 pieces,
    generic piece durations, the full callback.  Against the oracle; and the three ways the library can run it -- launch
    chain, fused tail with its state in global memory, fused tail with wave-local state (the default here) -- agree bit
    for bit on every per-point result and on the callback's value and gradient."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.reference_case(name, N=24)
    assert 60 <= len(w["points"]) <= 1000
    kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], poly_params=w["poly_params"],
              head_state=w["head_state"], tail_state=w["tail_state"])
    o = orc.Oracle(name, **kw)
    x = w["xs"][1]
    fo, go, _ = o.cost_function(w["points"], x, nthreads=NT)
    got = {}
    for label, env in (("default", {}), ("tail, global state", {"SVSDF_TAIL_LOCAL": "0"}), ("chain", {"SVSDF_TAIL": "off"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = svsdf_amd.SvsdfContext(shape=name, device=0, **kw)
        for k in env:
            monkeypatch.delenv(k)
        c.set_points(w["points"])
        for xx in w["xs"]:          # (the plan settles over the first callbacks: bound mode, anchor trial, tail)
            c.lmbm_evaluate(xx)
        f, g = c.lmbm_evaluate(x)
        st = c.stats()
        assert st["piece_time_exact"] == 1
        assert (st["tail_iter"] == 0) == (label != "chain"), (label, st["tail_iter"])
        coeffs, T = c.lmbm_prepare(x)
        got[label] = (f, g, c.query_points(coeffs, T)[:3])
        c.close()
    f, g, _ = got["default"]
    assert abs(f - fo) <= 1e-7 * abs(fo) and _rel(g, go) <= 1e-5, (f, fo, _rel(g, go))
    for label in ("tail, global state", "chain"):
        assert got[label][0] == f and np.array_equal(got[label][1], g), label
        for a, b in zip(got[label][2], got["default"][2]):
            assert np.array_equal(a, b), label


def test_group_stripes_serial_mode_and_fanout(built):
    """svsdf_group_stripe / svsdf_set_group_serial: three stripes on device 0.  The stripes partition the cloud (sizes within
    one point), every stripe reports its own counters and plan, the serial mode changes no bit, the fan-out time is
    reported and small."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C2", P=30000, minco=svsdf_amd.minco_coeffs)
    g = _ctx(w, devices=[0, 0, 0])
    g.set_points(w["points"])
    ref = None
    for _ in range(3):
        ref = g.eval_penalty(w["coeffs"], w["T"])
    st = g.stats()
    assert st["n_devices"] == 3 and 0.0 < st["fanout_ms"] < 5.0 and st["shader_clock_mhz"] > 500.0
    stripes = [g.group_stripe(k) for k in range(3)]
    assert sum(s["points"] for s in stripes) == 30000 and max(s["points"] for s in stripes) - min(s["points"] for s in stripes) <= 1
    assert all(s["device"] == 0 and s["stats"]["points"] == s["points"] and s["stats"]["solves"] > 0 for s in stripes)
    assert sum(s["stats"]["interior_points"] for s in stripes) == st["interior_points"]
    assert len({(s["plan"]["bound_mode"], s["plan"]["batches"], s["plan"]["lanes_per_query"]) for s in stripes}) == 1
    with pytest.raises(svsdf_amd.SvsdfError):
        g.group_stripe(3)
    g.set_group_serial(True)
    g.set_profiling(True)
    ser = g.eval_penalty(w["coeffs"], w["T"])
    assert ser[0] == ref[0] and np.array_equal(ser[1], ref[1]) and np.array_equal(ser[2], ref[2])
    dev = [g.group_stripe(k)["stats"]["device_ms"] for k in range(3)]
    assert all(d > 0.0 for d in dev) and max(dev) / (sum(dev) / 3) < 1.5      # striped Morton order: balanced work
    g.set_profiling(False)
    g.set_group_serial(False)
    again = g.eval_penalty(w["coeffs"], w["T"])
    assert again[0] == ref[0] and np.array_equal(again[2], ref[2])
    g.close()
    single = _ctx(w)
    single.set_points(w["points"])
    single.eval_penalty(w["coeffs"], w["T"])
    s0 = single.group_stripe(0)
    assert s0["points"] == 30000 and s0["stats"]["points"] == 30000
    with pytest.raises(svsdf_amd.SvsdfError):
        single.set_group_serial(True)
    single.close()


def test_contexts_share_the_stream_pool(built):
    """A context gives its streams back to the process-wide pool; the next context alone on the device gets the same set
    (round 5: destroying and re-creating streams cost the next context 7 % of its batch overlap).  Visible from outside as:
    create / destroy many contexts without the process's stream count growing -- here through results that stay
    bit-identical across 12 generations and two contexts alive at once."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C2", P=90000, minco=svsdf_amd.minco_coeffs)     # 3 batches: the batch streams are in use
    ref = None
    for gen in range(12):
        c = _ctx(w)
        c.set_points(w["points"])
        for _ in range(3):
            out = c.eval_penalty(w["coeffs"], w["T"])
        assert c.stats()["batches"] == 3
        if ref is None:
            ref = out
        assert out[0] == ref[0] and np.array_equal(out[2], ref[2]), gen
        if gen == 5:
            d = _ctx(w)                  # a second context while the first is alive: its own stream set
            d.set_points(w["points"][:5000])
            d.eval_penalty(w["coeffs"], w["T"])
            d.close()
        c.close()


def test_shape_selfcheck_and_what_it_guards(built, monkeypatch):
    """Every registered shape passes the 1-Lipschitz self-check taken at svsdf_create (the value-based second cull and the
    anchor bound mode are exact only for such an SDF).  A context told to distrust its shape runs without
    both -- same bits, fewer culled points."""
    import svsdf_amd
    from svsdf_amd import workload
    for shape in orc.SHAPES:
        poly = workload.mesh_outline("sdArc") if shape == "Polygon" else None
        c = svsdf_amd.SvsdfContext(shape=shape, device=0, polygon=poly, poly_params=(0.3, -0.2, 25.0) if shape != "Polygon" else (0, 0, 0))
        R, Rs, lip = c.shape_selfcheck()
        assert Rs <= R and lip == 0.0, (shape, R, Rs, lip)
        c.close()
    w = workload.make("C3", P=20000, minco=svsdf_amd.minco_coeffs)
    c = _ctx(w)
    c.set_points(w["points"])
    for _ in range(3):
        a = c.eval_penalty(w["coeffs"], w["T"])
    culled = c.stats()["culled_points"]
    c.close()
    monkeypatch.setenv("SVSDF_ASSUME_NOT_LIPSCHITZ", "1")
    d = _ctx(w)
    monkeypatch.delenv("SVSDF_ASSUME_NOT_LIPSCHITZ")
    assert d.shape_selfcheck()[2] > 0.0
    d.set_points(w["points"])
    for _ in range(5):
        b = d.eval_penalty(w["coeffs"], w["T"])
    assert d.stats()["gsip_bound_mode"] != 3 and d.stats()["culled_points"] < culled
    assert b[0] == a[0] and np.array_equal(b[1], a[1]) and np.array_equal(b[2], a[2])
    d.close()


def test_batches_rule_and_clock(built):
    """3 concurrent point batches from 80 k points per device in every bound mode (round 5), 1 below; the evaluation's
    shader clock as the kernel measures it is a plausible number."""
    import svsdf_amd
    from svsdf_amd import workload
    for P, nb in ((100000, 3), (60000, 1)):
        w = workload.make("C2", P=P, minco=svsdf_amd.minco_coeffs)
        c = _ctx(w)
        c.set_points(w["points"])
        for _ in range(3):
            c.eval_penalty(w["coeffs"], w["T"])
        st = c.stats()
        assert st["batches"] == nb and c.get_plan()["batches"] == nb, (P, st["batches"])
        assert 800.0 < st["shader_clock_mhz"] < 3500.0, st["shader_clock_mhz"]
        c.close()


@pytest.mark.parametrize("N,piece_s", [(65, 1.7), (96, 1.2), (128, 0.9)])
def test_more_than_64_pieces(built, N, piece_s):
    """SVSDF_MAX_PIECES is 128 since round 5 (64 before; the reference has no cap, minco.hpp:433-513): trajectories of 65 /
    96 / 128 pieces with generic durations -- the penalty, every per-point result and the full optimizer callback (host MINCO
    forward + adjoint at that N) against the oracle; one piece more than the cap is refused with SVSDF_ERR_INVALID."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C3", P=4000, N=N)
    rng = np.random.default_rng(N)
    w["T"] = piece_s * (1.0 + 1e-3 * rng.standard_normal(N))
    w["coeffs"] = svsdf_amd.minco_coeffs(w["head_state"], w["tail_state"], w["q"], w["T"])
    c = _ctx(w)
    c.set_points(w["points"])
    o = _oracle(w)
    o.set_traj(w["coeffs"], w["T"])
    ocost, ogT, ogC, osdf, ots, _ = o.penalty(w["points"], nthreads=NT, sum_mode=1, per_point=True)
    for _ in range(3):
        cost, gT, gC = c.eval_penalty(w["coeffs"], w["T"])
    sdf, ts, g, _ = c.query_points(w["coeffs"], w["T"])
    assert ocost > 0 and int((osdf <= 0).sum()) > 50
    flips = np.abs(ts - ots) > 1e-6
    assert flips.mean() <= 5e-3, int(flips.sum())
    assert np.abs(sdf[~flips] - osdf[~flips]).max() <= 1e-7
    assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
    assert _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5, (_rel(gC, ogC), _rel(gT, ogT))
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
    f, gx = c.lmbm_evaluate(x)
    fo, go, c3 = o.cost_function(w["points"], x, nthreads=NT)
    assert abs(f - fo) <= 1e-7 * abs(fo), (f, fo)
    assert _rel(gx, go) <= 1e-5, _rel(gx, go)
    if N == 128:
        w2 = workload.make("C3", P=100, N=129)
        T2 = np.full(129, 0.9)
        co2 = svsdf_amd.minco_coeffs(w2["head_state"], w2["tail_state"], w2["q"], T2)
        with pytest.raises(Exception):
            c.eval_penalty(co2, T2)
        cost3, _, _ = c.eval_penalty(w["coeffs"], w["T"])      # the context is still usable
        assert cost3 == cost
    c.close()


def test_small_cloud_lazy_trial(built):
    """Round 5: a small cloud (the fused tail owns the GSIP loop) of a cheap-bound shape TRIES the lazy scans in its second
    evaluation and keeps them when they save a quarter of the GSIP solves -- decided by counters, so two contexts agree, the
    plan settles within four evaluations, and no bit of the result depends on it.  16 pieces: kept; 8 pieces (C1): not."""
    import svsdf_amd
    from svsdf_amd import workload
    for config, P, want in (("C2", 3000, 2), ("C1", 10000, 0)):   # (C1 at its BASELINE size: 86 % of the solves left -> rejected)
        w = workload.make(config, P=P, minco=svsdf_amd.minco_coeffs)
        plans, vals = [], []
        for _ in range(2):
            c = _ctx(w)
            c.set_points(w["points"])
            n = 0
            first = None
            while n < 8 and not c.get_plan()["settled"]:
                out = c.eval_penalty(w["coeffs"], w["T"])
                if first is None:
                    first = out
                assert out[0] == first[0] and np.array_equal(out[2], first[2]), (config, n)   # trial and final mode: same bits
                n += 1
            assert n <= 4, n
            plans.append((c.get_plan(), n))
            vals.append(c.eval_penalty(w["coeffs"], w["T"]))
            assert c.stats()["tail_iter"] == 0
            c.close()
        assert plans[0] == plans[1], plans
        assert plans[0][0]["bound_mode"] == want, (config, plans[0])
        p = _ctx(w)
        p.set_points(w["points"])
        p.set_plan(bound_mode=0)
        ref = p.eval_penalty(w["coeffs"], w["T"])
        assert ref[0] == vals[0][0] and np.array_equal(ref[1], vals[0][1]) and np.array_equal(ref[2], vals[0][2])
        p.close()
