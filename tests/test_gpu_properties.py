"""GPU tests at BASELINE.json's full sizes through size-independent properties (the oracle would
need minutes-hours there): determinism, solver-variant bit-identity (exact pruning, lane groups),
shard additivity, exact-zero contribution of far points, accumulate semantics."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(w, **kw):
    import svsdf_amd
    return svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"],
                                  rho=w["rho"], poly_params=w["poly_params"], polygon=w["polygon"],
                                  head_state=w["head_state"], tail_state=w["tail_state"], device=0, **kw)


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def c2(built):
    import svsdf_amd
    from svsdf_amd import workload
    return workload.make("C2", minco=svsdf_amd.minco_coeffs)   # 100k points, star, N = 16


def test_c2_full_size_solver_variants_bit_identical(c2):
    """Pruned layer-1 scan == full scan and every lane-group width / batch split give the same
    (sdf, t*, grad) for all 100k points (the restructurings are exact, not approximate)."""
    def run(env):
        def go():
            c = _ctx(c2)
            c.set_points(c2["points"])
            out = c.query_points(c2["coeffs"], c2["T"])
            st = c.stats()
            c.close()
            return out, st
        return _with_env(env, go)
    # reference configuration: full scan, one lane per query, every GSIP sample solved
    ref, st_ref = run(dict(SVSDF_G=1, SVSDF_G_LATE=1, SVSDF_PRUNE=0, SVSDF_BATCHES=1, SVSDF_SELECT_DELTA=1e9))
    assert st_ref["solves"] == st_ref["gsip_samples"] + 100000
    assert st_ref["scan_evals"] == st_ref["solves"] * 267          # K = floor(40 / 0.15) + 1
    for env in (dict(SVSDF_G=4, SVSDF_G_LATE=8, SVSDF_PRUNE=1, SVSDF_BATCHES=1),
                dict(SVSDF_G=2, SVSDF_G_LATE=2, SVSDF_PRUNE=1, SVSDF_BATCHES=4),
                dict(SVSDF_G=8, SVSDF_G_LATE=8, SVSDF_PRUNE=1, SVSDF_BATCHES=3),
                dict(SVSDF_G=16, SVSDF_G_LATE=32, SVSDF_PRUNE=1, SVSDF_BATCHES=1),
                dict(SVSDF_G=32, SVSDF_G_LATE=16, SVSDF_PRUNE=1, SVSDF_BATCHES=2)):
        out, st = run(env)
        for a, b in zip(out, ref):
            assert np.array_equal(a, b), env
        assert st["gsip_samples"] == st_ref["gsip_samples"] and st["interior_points"] == st_ref["interior_points"]
        assert st["scan_evals"] < 0.2 * st_ref["scan_evals"]
        assert st["solves"] < 0.6 * st_ref["solves"]


def test_c2_full_size_determinism_and_shard_additivity(c2):
    import svsdf_amd
    c = _ctx(c2)
    c.set_points(c2["points"])
    a = c.query_points(c2["coeffs"], c2["T"])
    b = c.query_points(c2["coeffs"], c2["T"])
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    full = c.eval_penalty(c2["coeffs"], c2["T"])
    acc = (0.0, np.zeros_like(full[1]), np.zeros_like(full[2]))
    n = 0
    for r in range(4):
        s = _ctx(c2, rank=r, world_size=4)
        s.set_points(c2["points"])
        n += s.num_points()
        acc = s.eval_penalty(c2["coeffs"], c2["T"], *acc)
        s.close()
    assert n == len(c2["points"])
    assert abs(acc[0] - full[0]) <= 1e-11 * abs(full[0])
    np.testing.assert_allclose(acc[2], full[2], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(acc[1], full[1], rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize("config", ["C3", "C5"])
def test_million_points_far_points_contribute_exact_zero(built, config):
    """1M points (sdHorseshoe N=32 / Polygon N=16): appending 200k points farther than
    R_shape + safety_hor from the whole path must not change a single bit of the result of a
    context holding only the original points evaluated in the same order (exact-zero rule of
    smoothedL1, BEO:321-324) -- here checked as cost equality and gradient closeness, plus
    the half/half split adding up."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(config, P=1_000_000, minco=svsdf_amd.minco_coeffs)
    c = _ctx(w)
    c.set_points(w["points"])
    full = c.eval_penalty(w["coeffs"], w["T"])
    st = c.stats()
    assert st["points"] == 1_000_000 and full[0] > 0
    rng = np.random.default_rng(3)
    far = np.zeros((200_000, 3))
    far[:, 0] = rng.uniform(200.0, 400.0, len(far))
    far[:, 1] = rng.uniform(-300.0, 300.0, len(far))
    c.set_points(np.concatenate([w["points"], far]))
    both = c.eval_penalty(w["coeffs"], w["T"])
    assert abs(both[0] - full[0]) <= 1e-12 * abs(full[0])
    np.testing.assert_allclose(both[2], full[2], rtol=1e-10, atol=1e-6)
    c.set_points(far)
    z = c.eval_penalty(w["coeffs"], w["T"])
    assert z[0] == 0.0 and not z[1].any() and not z[2].any()
    c.close()


def test_inlined_sincos_is_bit_identical_to_device_library(built):
    """The kernels inline the ROCm device library's sincos arithmetic (small-argument path);
    every result bit must agree with the library routine, including arguments around the
    2^30 hand-over and negative / tiny / zero arguments."""
    import svsdf_amd
    c = svsdf_amd.SvsdfContext(shape="star", device=0)
    for lo, hi, n in ((-10.0, 10.0, 4_000_001), (-1e-300, 1e-300, 1001), (-4000.0, 4000.0, 2_000_001),
                      (1.0e9, 1.2e9, 100_001), (-3.0e10, 3.0e10, 100_001), (0.0, 0.0 + 1e-12, 1001)):
        assert c.sincos_mismatches(lo, hi, n) == 0, (lo, hi)
    c.close()


def test_exact_cull_is_invisible(built):
    """The exact cull (points proven inactive from the chunk bounds -- and, second stage, from the scanned table values --
    plus a rigorous continuous-path allowance skip the argmin solve) must not change cost or gradient, and may only ever drop points whose true
    SVSDF exceeds safety_hor (checked against the un-culled per-point query)."""
    import svsdf_amd
    from svsdf_amd import workload
    for cfg, P in (("C2", 60000), ("C4", 40000), ("C5", 30000)):
        w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)

        def run():
            ctx = _ctx(w)
            ctx.set_points(w["points"])
            out = ctx.eval_penalty(w["coeffs"], w["T"])
            st = ctx.stats()
            sdf = ctx.query_points(w["coeffs"], w["T"])[0]   # query_points never culls
            return out, st, sdf
        (c0, gT0, gC0), st0, sdf0 = _with_env(dict(SVSDF_CULL=0), run)
        inactive = int((sdf0 > w["safety_hor"]).sum())
        prev = 0
        for level in (1, 2):    # 1: the chunks' bounding circles; 2 (default): also the scanned table values (round 4)
            (c1, gT1, gC1), st1, sdf1 = _with_env(dict(SVSDF_CULL=level), run)
            assert st0["culled_points"] == 0 and prev < st1["culled_points"] <= inactive
            assert st1["solves"] == st0["solves"] - st1["culled_points"]
            np.testing.assert_array_equal(sdf0, sdf1)
            assert c1 == c0                                          # same non-zero terms, deterministic assembly
            np.testing.assert_array_equal(gT1, gT0)
            np.testing.assert_array_equal(gC1, gC0)
            prev = st1["culled_points"]
    # stale-duration regime (total >= 300 s after a shorter trajectory): the cull switches itself off
    w = workload.make("C2", P=20000, minco=svsdf_amd.minco_coeffs)
    ctx = _ctx(w)
    ctx.set_points(w["points"])
    ctx.eval_penalty(w["coeffs"], w["T"])
    assert ctx.stats()["culled_points"] > 0
    Tlong = np.asarray(w["T"]) * 8.0                            # 320 s
    ctx.eval_penalty(svsdf_amd.minco_coeffs(w["head_state"], w["tail_state"], w["q"], Tlong), Tlong)
    assert ctx.stats()["culled_points"] == 0


def test_gsip_bound_modes_are_invisible(built):
    """The two qualities of GSIP upper bound (nearest-chunk bound vs the sample's own layer-1 scan done by k_round
    and reused by k_solve) only decide WHICH samples are solved first: per-point results, cost and
    gradients must be identical, and the full mode must need fewer solves where the cheap bound is poor."""
    import svsdf_amd
    from svsdf_amd import workload
    for cfg, P in (("C2", 40000), ("C3", 60000), ("C4", 40000), ("C5", 20000)):
        w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)

        def run():
            ctx = _ctx(w)
            ctx.set_points(w["points"])
            out = ctx.eval_penalty(w["coeffs"], w["T"])
            st = ctx.stats()
            q = ctx.query_points(w["coeffs"], w["T"])
            return out, st, q
        (c0, gT0, gC0), st0, q0 = _with_env(dict(SVSDF_UB_FULL=0), run)
        (c1, gT1, gC1), st1, q1 = _with_env(dict(SVSDF_UB_FULL=1), run)
        (c2, gT2, gC2), st2, q2 = _with_env(dict(SVSDF_UB_FULL=2), run)     # lazy: scans only the cheap bound's band
        for a, b in zip(q0[:3], q2[:3]):
            np.testing.assert_array_equal(a, b)
        assert c2 == c0
        np.testing.assert_array_equal(gC2, gC0)
        assert st2["solves"] <= st0["solves"] and st2["gsip_bound_mode"] == 2
        for a, b in zip(q0[:3], q1[:3]):
            np.testing.assert_array_equal(a, b)          # sdf, t*, gradient direction: bit for bit
        assert c1 == c0
        np.testing.assert_array_equal(gC1, gC0)
        np.testing.assert_array_equal(gT1, gT0)
        assert st1["solves"] <= st0["solves"], (cfg, st0["solves"], st1["solves"])
        if cfg == "C3":
            assert st1["solves"] < 0.6 * st0["solves"]




def test_interior_capacity_grows_and_repeats(built, monkeypatch):
    """The GSIP arrays are sized by the interior count, not by the cloud (round 4): an evaluation that finds more interior
    points than the arrays hold drops the surplus, the library grows the arrays and repeats the evaluation.  Forced here
    with a 64-entry start capacity: cost, gradients and every per-point result must equal the normal run bit for bit, also
    after the trajectory moved (more interior points than the fitted capacity)."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C2", P=20000, minco=svsdf_amd.minco_coeffs)
    kw = dict(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], head_state=w["head_state"],
              tail_state=w["tail_state"], device=0)
    ref = svsdf_amd.SvsdfContext(**kw)
    ref.set_points(w["points"])
    monkeypatch.setenv("SVSDF_ICAP_INIT", "64")
    small = svsdf_amd.SvsdfContext(**kw)
    small.set_points(w["points"])
    monkeypatch.delenv("SVSDF_ICAP_INIT")
    wide = dict(kw, safety_hor=w["safety_hor"])
    for k, scale in enumerate((1.0, 1.0, 0.97)):      # the third trajectory is another one: the interior set changes
        q = w["q"] * scale
        coeffs = svsdf_amd.minco_coeffs(w["head_state"], w["tail_state"], q, w["T"])
        a = ref.eval_penalty(coeffs, w["T"])
        b = small.eval_penalty(coeffs, w["T"])
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), k
        qa, qb = ref.query_points(coeffs, w["T"]), small.query_points(coeffs, w["T"])
        for x, y in zip(qa[:3], qb[:3]):
            assert np.array_equal(x, y), k
    assert small.stats()["interior_points"] == ref.stats()["interior_points"] > 64
