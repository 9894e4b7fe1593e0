"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (north_star): gradient vector <= 1e-5 relative (L2); cost <= 1e-7 relative here (the
per-point argmin is iterative, so t* agrees to ~1e-8 and the cost to ~1e-9).  Per-point basin
flips (|t*_hip - t*_oracle| > 1e-6) are counted and must stay rare.
"""
import os

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
NT = os.cpu_count() or 1


def _mk(config, P, N=None, dist="corridor", seed=None):
    import svsdf_amd
    from svsdf_amd import workload
    kw = {} if seed is None else {"seed": seed}
    w = workload.make(config, P=P, N=N, dist=dist, minco=svsdf_amd.minco_coeffs, **kw)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"],
                                 rho=w["rho"], poly_params=w["poly_params"], polygon=w["polygon"],
                                 head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                   poly_params=w["poly_params"], polygon=w["polygon"],
                   head_state=w["head_state"], tail_state=w["tail_state"])
    o.set_traj(w["coeffs"], w["T"])
    return w, ctx, o


def _rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


@pytest.mark.parametrize("config,P", [("C1", 3000), ("C2", 2000)])
def test_per_point_queries_match_oracle(built, config, P):
    w, ctx, o = _mk(config, P)
    sdf, ts, g, idx = ctx.query_points(w["coeffs"], w["T"])
    assert np.array_equal(idx, np.arange(P))
    osdf, ots, og = o.query(w["points"], nthreads=NT)
    flips = np.abs(ts - ots) > 1e-6
    assert flips.mean() <= 2e-3, flips.sum()
    ok = ~flips
    ext = ok & (osdf > 0)
    np.testing.assert_allclose(sdf[ext], osdf[ext], rtol=0, atol=1e-9)
    np.testing.assert_allclose(g[ext], og[ext], rtol=0, atol=2e-6)   # FD gradient, dx = 1e-6
    itr = ok & (osdf <= 0)
    assert itr.sum() > 0.1 * P
    np.testing.assert_allclose(sdf[itr], osdf[itr], rtol=0, atol=1e-7)
    np.testing.assert_allclose(g[itr], og[itr], rtol=0, atol=1e-6)
    st = ctx.stats()
    cnt = o.counters()
    assert st["interior_points"] == cnt["interior_points"]
    # the reference solves every GSIP circle sample; the upper-bound selection solves a subset
    assert st["gsip_samples"] + P == cnt["solves"]
    assert P < st["solves"] <= cnt["solves"]


@pytest.mark.parametrize("config,P,N", [("C1", 4000, None), ("C2", 3000, None), ("C3", 2000, None),
                                        ("C4", 2000, None), ("C5", 1000, 8)])
def test_penalty_matches_oracle(built, config, P, N):
    w, ctx, o = _mk(config, P, N=N)
    Np = len(w["T"])
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    ocost, ogT, ogC = o.penalty(w["points"], nthreads=NT, sum_mode=1)
    assert ocost > 0
    assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
    assert _rel(gC, ogC) <= 1e-5, _rel(gC, ogC)
    assert _rel(gT, ogT) <= 1e-5, _rel(gT, ogT)
    # accumulate-into semantics (BEO:774-779: += on cost / gradT / gradC)
    c0, t0, C0 = 3.5, np.arange(Np, dtype=float), np.ones((6 * Np, 3))
    cost2, gT2, gC2 = ctx.eval_penalty(w["coeffs"], w["T"], c0, t0, C0)
    assert abs((cost2 - c0) - cost) <= 1e-9 * abs(cost)
    np.testing.assert_allclose(gT2 - t0, gT, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gC2 - C0, gC, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("config,P", [("C1", 3000), ("C2", 1500), ("C3", 1000)])
def test_full_callback_matches_oracle(built, config, P):
    import svsdf_amd
    from svsdf_amd import workload
    w, ctx, o = _mk(config, P)
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
    # perturb so that tau/T are not all equal
    rng = np.random.default_rng(5)
    x = x + 0.05 * rng.standard_normal(len(x))
    f, g = ctx.lmbm_evaluate(x)
    fo, go, c3 = o.cost_function(w["points"], x, nthreads=NT)
    assert abs(f - fo) <= 1e-7 * abs(fo), (f, fo)
    assert _rel(g, go) <= 1e-5, _rel(g, go)
    np.testing.assert_allclose(ctx.last_costs(), c3, rtol=1e-7)


@pytest.mark.parametrize("shape", ["sdUnevenCapsule", "sdCutDisk", "sdTrapezoid", "sdRhombus", "sdTunnel",
                                   "sdOrientedVesica", "sdRoundedCross", "sdRoundedX", "bigX", "sdMoon",
                                   "sdPie", "sdPie2", "sdArc", "Polygon"])
def test_every_shape_matches_oracle(built, shape):
    """All 16 registered shapes + the fallback Polygon rectangle (SWM:363-369)."""
    import svsdf_amd
    from svsdf_amd import workload
    pp = (0.0, -3.0, 0.0) if shape == "sdCutDisk" else ((0.3, -0.2, 25.0) if shape == "sdPie" else (0.0, 0.0, 0.0))
    base = workload.make("C1", P=600, minco=svsdf_amd.minco_coeffs)
    kw = dict(safety_hor=0.7, weight_p=60.0, rho=3.8, poly_params=pp,
              head_state=base["head_state"], tail_state=base["tail_state"])
    ctx = svsdf_amd.SvsdfContext(shape=shape, device=0, **kw)
    ctx.set_points(base["points"])
    o = orc.Oracle(shape, **kw)
    o.set_traj(base["coeffs"], base["T"])
    cost, gT, gC = ctx.eval_penalty(base["coeffs"], base["T"])
    ocost, ogT, ogC = o.penalty(base["points"], nthreads=NT, sum_mode=1)
    assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
    assert _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5


def test_edge_cases(built):
    import svsdf_amd
    from svsdf_amd import workload
    w, ctx, o = _mk("C1", 64)
    # single point, far away: contributes exactly zero
    ctx.set_points(np.array([[500.0, 500.0, 3.0]]))
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    assert cost == 0.0 and not gT.any() and not gC.any()
    # z is ignored (BEO:790-791)
    p = w["points"][:32].copy()
    ctx.set_points(p)
    a = ctx.eval_penalty(w["coeffs"], w["T"])
    p[:, 2] = 7.0
    ctx.set_points(p)
    b = ctx.eval_penalty(w["coeffs"], w["T"])
    assert a[0] == b[0] and np.array_equal(a[2], b[2])
    # evaluate before set_points / bad n
    c2 = svsdf_amd.SvsdfContext(shape="star", device=0)
    with pytest.raises(svsdf_amd.SvsdfError):
        c2.eval_penalty(w["coeffs"], w["T"])
    with pytest.raises(svsdf_amd.SvsdfError):
        ctx.lmbm_evaluate(np.zeros(6))
    # N = 1 (no interior waypoints) works
    hs, ts = w["head_state"], w["tail_state"]
    c1 = svsdf_amd.minco_coeffs(hs, ts, np.zeros((0, 3)), np.array([30.0]))
    ctx.set_points(w["points"])
    cost1, _, _ = ctx.eval_penalty(c1, np.array([30.0]))
    o.set_traj(c1, np.array([30.0]))
    oc1, _, _ = o.penalty(w["points"], nthreads=NT)
    assert abs(cost1 - oc1) <= 1e-7 * max(abs(oc1), 1.0)


def test_sharded_contexts_sum_to_whole(built):
    """world_size-2 sharding in one process: the two shard partials add up to the full result."""
    import svsdf_amd
    w, ctx, o = _mk("C1", 2000)
    full = ctx.eval_penalty(w["coeffs"], w["T"])
    acc_c, acc_T, acc_C = 0.0, np.zeros_like(full[1]), np.zeros_like(full[2])
    seen = []
    for r in range(2):
        c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"],
                                   rho=w["rho"], head_state=w["head_state"], tail_state=w["tail_state"],
                                   device=0, rank=r, world_size=2)
        c.set_points(w["points"])
        seen.append(c.shard_indices())
        acc_c, acc_T, acc_C = c.eval_penalty(w["coeffs"], w["T"], acc_c, acc_T, acc_C)
    assert sorted(np.concatenate(seen).tolist()) == list(range(2000))
    assert abs(acc_c - full[0]) <= 1e-10 * abs(full[0])
    np.testing.assert_allclose(acc_C, full[2], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(acc_T, full[1], rtol=1e-9, atol=1e-9)


def _budget_points(evals_per_point, seconds=8.0):
    """Points the oracle finishes in ~`seconds` on this host (~1.2e6 SDF evaluations/s per core, measured)."""
    return int(seconds * 1.2e6 * NT / evals_per_point)


@pytest.mark.parametrize("config,full,evals_pp", [("C2", 100000, 6500), ("C3", 1000000, 11000), ("NS", 1000000, 7600),
                                                  ("C4", 500000, 11000), ("C5", 100000, 90000)])
def test_baseline_sizes_match_oracle(built, config, full, evals_pp):
    """The BASELINE.json workloads against the oracle of record, EVERY point compared: per-point SVSDF and t*, interior
    count, reduced cost and gradients.  On a host with >= 128 threads (the GPU boxes have 256) the sizes are the full
    ones -- all 100 k points of C2, all 1 M of C3 and of the north-star workload (~ 25 s of oracle each), 100 k of C5 (its
    77-vertex outline costs the oracle ~ 2 minutes per 100 k points on 256 threads; round 5 had halved it, round 6 restored
    it: the suite uses a third of its 1 200 s step), 500 k of C4 (the other half of a device's share of
    its 4 M is the same distribution); smaller hosts compare what the oracle finishes in ~ 8 s.
    Round 3 one-off at these sizes (256 cores): basin flips 11 / 22 / 105 / 5 / 3, cost rel <= 5e-13, gradC rel <= 2.4e-7."""
    P = full if NT >= 128 else max(2000, min(full, _budget_points(evals_pp)))
    w, ctx, o = _mk(config, P)
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    ocost, ogT, ogC, osdf, ots, _ = o.penalty(w["points"], nthreads=NT, sum_mode=1, per_point=True)
    flips = np.abs(ts - ots) > 1e-6
    assert flips.mean() <= 2e-3, (int(flips.sum()), P)
    assert np.abs(sdf[~flips] - osdf[~flips]).max() <= 1e-7
    assert ctx.stats()["interior_points"] == o.counters()["interior_points"]
    assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
    assert _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5, (_rel(gC, ogC), _rel(gT, ogT))
    print(f"{config}: P = {P} (all points compared), flips {int(flips.sum())}, cost rel {abs(cost - ocost) / abs(ocost):.2e}, "
          f"gradC rel {_rel(gC, ogC):.2e}, gradT rel {_rel(gT, ogT):.2e}")


@pytest.mark.parametrize("config,evals_pp", [("C2", 3000), ("C3", 5000), ("C4", 5000)])
def test_map_distribution_matches_oracle(built, config, evals_pp):
    """Second point distribution of SURVEY.md §8(d): uniform over the demo map extent [0, 30.5] x [0, 75] m (most
    points far from the path: the exact cull carries the evaluation).  Same gates as the corridor cloud."""
    P = max(4000, min(200000, _budget_points(evals_pp)))
    w, ctx, o = _mk(config, P, dist="map")
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    st = ctx.stats()
    ocost, ogT, ogC, osdf, ots, _ = o.penalty(w["points"], nthreads=NT, sum_mode=1, per_point=True)
    flips = np.abs(ts - ots) > 1e-6
    assert flips.mean() <= 2e-3, (int(flips.sum()), P)
    assert np.abs(sdf[~flips] - osdf[~flips]).max() <= 1e-7
    assert st["interior_points"] == o.counters()["interior_points"]
    assert st["culled_points"] > 0.3 * P
    assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
    assert _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5, (_rel(gC, ogC), _rel(gT, ogT))
    print(f"{config} map: P = {P}, flips {int(flips.sum())}, culled {st['culled_points']}, cost rel "
          f"{abs(cost - ocost) / abs(ocost):.2e}, gradC rel {_rel(gC, ogC):.2e}, gradT rel {_rel(gT, ogT):.2e}")


def _source_seed():
    """A fuzz seed nobody picked: derived from the commit when .git is there, else (the GPU box gets a snapshot without
    .git) from a hash of the product sources -- it moves with every change of the kernels."""
    import hashlib, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        h = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           check=True).stdout.decode().strip()
        if h:
            return int(h[:8], 16) % 1000003
    except Exception:
        pass
    m = hashlib.sha256()
    src = os.path.join(root, "implicit-svsdf-planner_amd", "csrc")
    for f in sorted(os.listdir(src)):
        m.update(open(os.path.join(src, f), "rb").read())
    return int(m.hexdigest()[:8], 16) % 1000003


def _fuzz(cases, seed, **env):
    import ast, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), str(cases), str(seed)],
                         env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True).stdout.decode()
    last = out.strip().splitlines()[-1]
    return ast.literal_eval(last[last.index("worst") + 6:last.rindex("}") + 1]), out


def test_differential_fuzz(built):
    """tools/fuzz_parity.py: random (shape incl. mesh outlines, shape offset, 1-6 piece trajectory with generic
    durations, safety margin, 400 points) cases WITH the degenerate points (exactly on waypoints = on the zero level set
    of some shapes at a rest pose), HIP vs the oracle of record (glibc trig, reference piece location), three fixed
    seeds (a fourth, derived from the commit / the source tree, when SVSDF_FUZZ_NIGHTLY=1: it changes with every commit,
    so it is not part of the blocking set; the seed is printed).  Gates: cost 1e-7, gradient 1e-5 (north_star), basin
    flips 1 %.  A case outside them is admitted ONLY through the sensitivity bracket (round 4): the oracle of record
    re-run with its sin / cos / atan2 results moved by <= 1 ulp (SEVEN seeds -- up to 28 for a case that is bit-identical to the device-trig oracle and misses the bracket --, round 4: three; no device-library arithmetic
    involved) must itself move by at least HALF the HIP deviation on every violated metric (round 4: a quarter).  VERDICT
    r4's "ratio >= 1, ceilings 1e-6 / 1e-3 / 6 %" was run first and rejects the reference against itself (tools/fuzz_parity.py
    header, profiles/r05_fuzz_tight_first_attempt.txt).  Round 6: the bracket is no longer sufficient on its own --
    an admitted case must ALSO be bit-identical per point (SVSDF value, t*) to the oracle evaluated with the device library's
    trig, so that its whole deviation is the trig difference the bracket prices.  The admitted fraction goes to the test log as a warning (visible
    under -q).  These are plateaus of SDF(t) -- resting end poses,
    turn-on-the-spot trajectories -- where the reference's own result depends on the libm it was built with
    (tests/test_plateau_sensitivity.py).  Everything else fails the test."""
    seeds = [7, 20240807, 424243]
    if os.environ.get("SVSDF_FUZZ_NIGHTLY") == "1":
        seeds.append(_source_seed())
    total_explained = 0
    for seed in seeds:
        worst, out = _fuzz(40, seed, FUZZ_DEGENERATE="1")
        assert "HIP error" not in out, out[-2000:]
        print(f"seed {seed}: worst {worst}")
        assert worst["unexplained"] == 0, out[-4000:]
        total_explained += worst["libm_explained"]
    msg = (f"differential fuzz: {total_explained} of {40 * len(seeds)} cases ({100.0 * total_explained / (40 * len(seeds)):.1f} %) outside the "
           f"gates, all admitted through the 1-ulp bracket of the oracle (7 seeds or more, ratio >= 0.5) AND bit-identical per point to the device-trig oracle; 0 unexplained")
    print(msg)
    import warnings
    warnings.warn(msg)   # (shows in the -q summary: the GPUTEST tail carries the admitted fraction)
    assert total_explained <= 0.10 * 40 * len(seeds)   # plateaus are rare, not the rule (fresh 200-case campaigns: 1.5 % and 5 %)
    # the round-1 arithmetic (forced) on the first seed: inside the old, looser gate only
    worst_fast, out = _fuzz(40, 7, FUZZ_DEGENERATE="0", FUZZ_PIECE_TIME="fast")
    assert worst_fast["cost"] <= 1e-7 and worst_fast["gC"] <= 1e-4, out[-2000:]


@pytest.mark.parametrize("config,P", [("C1", 6000), ("C2", 6000), ("C3", 4000), ("C4", 4000), ("C5", 3000)])
def test_bit_identical_when_oracle_uses_device_trig(built, config, P):
    """The HIP kernels' arithmetic differs from the reference's, operation for operation, in exactly two places:
    (a) sin/cos (pose yaw, GSIP sample angles) and the one atan2 come from the ROCm device library instead of
    glibc, (b) the local time of piece i is t - (T_0 + ... + T_{i-1}) instead of i successive subtractions
    (TRJ:498-516; no difference for the equal 2.5 s pieces of the BASELINE configs).  With the oracle switched to
    the same two choices (orc_set_trig_mode(1): the device library's published algorithms written in C) the HIP
    pipeline -- pose table, exact pruning, lane groups, upper-bound sample selection, cull, all of it -- must
    reproduce the oracle's per-point SVSDF, t* and gradient direction BIT FOR BIT."""
    if config == "C2" and NT >= 64:
        P = 100000   # the whole BASELINE workload where the host can afford it (2 s on the 256-core GPU box)
    w, ctx, o = _mk(config, P)
    o.set_trig_mode(1)
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    osdf, ots, og = o.query(w["points"], nthreads=NT)
    same_t = ts == ots
    same_s = sdf == osdf
    same_g = (g == og).all(axis=1)
    print(f"{config}: identical t* {same_t.mean():.5f}  sdf {same_s.mean():.5f}  grad {same_g.mean():.5f} of {P}")
    assert same_t.all() and same_s.all() and same_g.all()


def test_differential_fuzz_bit_identical_in_device_arithmetic_mode(built):
    """Same kind of random cases as test_differential_fuzz (all 17 shapes incl. mesh outlines, shape offsets, 1-6 unequal
    pieces, degenerate points), with the oracle in device-arithmetic mode: not one of the per-point (t*, SVSDF) values
    may differ in any bit, and cost / gradients agree to summation order (1e-12).  Fixed seeds: 11 and 457738 -- the
    latter is the commit-derived seed that, in round 4, found the shape rotation one ulp off (cos / sin vs glibc's sincos,
    tests/test_gpu_sdf_at.py); a seed derived from the current commit is added with SVSDF_FUZZ_NIGHTLY=1 (it changes
    with every commit, so it is not part of the blocking set; it is printed)."""
    seeds = [11, 457738]
    if os.environ.get("SVSDF_FUZZ_NIGHTLY") == "1":
        seeds.append(_source_seed() + 1)
    for seed in seeds:
        worst, out = _fuzz(40, seed, FUZZ_DEGENERATE="1", FUZZ_DEVICE_TRIG="1")
        print(f"seed {seed}: {worst}")
        assert worst["not_identical"] == 0, out[-3000:]
        assert worst["cost"] <= 1e-12 and worst["gC"] <= 1e-12 and worst["gT"] <= 1e-12 and worst["flips"] == 0.0, out[-2000:]
