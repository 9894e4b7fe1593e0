"""svsdf_swept_outline (second half: what the reference's sw_calculate / calculateSwept are for) on the GPU,
checked against the CPU oracle: the outline is the zero set of getSDFofSweptVolume (vertices) and separates the points the
oracle's getTrueSDFofSweptVolume calls interior from the exterior ones (probes)."""
import os

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
NT = os.cpu_count() or 1


def _inside(loops, pts):
    """even-odd rule over all loops (outer boundaries and holes alike)"""
    inside = np.zeros(len(pts), dtype=bool)
    x, y = pts[:, 0], pts[:, 1]
    for lp in loops:
        x0, y0 = lp[:, 0], lp[:, 1]
        x1, y1 = np.roll(x0, -1), np.roll(y0, -1)
        for a, b, c, d in zip(x0, y0, x1, y1):
            crosses = ((b > y) != (d > y)) & (x < (c - a) * (y - b) / (d - b + 1e-300) + a)
            inside ^= crosses
    return inside


@pytest.mark.parametrize("config,cell", [("C1", 0.05), ("C2", 0.08)])
def test_swept_outline_is_the_oracles_zero_set(built, config, cell):
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(config, P=200, minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                                 tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    before = ctx.eval_penalty(w["coeffs"], w["T"])
    loops, st = ctx.swept_outline(w["coeffs"], w["T"], cell=cell)
    after = ctx.eval_penalty(w["coeffs"], w["T"])     # the caller's resident cloud and plan are untouched
    assert before[0] == after[0] and np.array_equal(before[1], after[1]) and np.array_equal(before[2], after[2])
    assert st["open_chains"] == 0 and len(loops) >= 1
    assert st["nodes_evaluated"] < 0.25 * st["dense_nodes"]
    areas = [0.5 * np.sum(lp[:, 0] * np.roll(lp[:, 1], -1) - np.roll(lp[:, 0], -1) * lp[:, 1]) for lp in loops]
    assert max(areas) > 0.0                            # the outer boundary runs counter-clockwise
    o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                   poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"])
    o.set_traj(w["coeffs"], w["T"])
    # 1. the outline vertices lie on the oracle's zero set up to the interpolation error of a cell.  (Not every one: the
    # reference's argmin search returns the local minimum next to its seed, so the marched function jumps where the
    # seed changes basin -- at a star's concave corners, at the path's ends -- and a crossing interpolated across such
    # a jump is off by the jump.)
    verts = np.vstack(loops)
    sdf = np.abs(np.array([o.sdf_swept(x, y)[0] for x, y in verts[::3]]))   # getSDFofSweptVolume: the marched function
    assert np.median(sdf) < 0.1 * cell and np.quantile(sdf, 0.97) < 0.25 * cell, (np.median(sdf), np.quantile(sdf, 0.97))
    # 2. the path itself is inside, points a shape diameter away are outside
    ts = np.linspace(0.0, float(np.sum(w["T"])), 60)
    path = np.array([o.pos(t)[:2] for t in ts])
    assert _inside(loops, path).all()
    # 3. inside / outside by the outline == sign of the oracle's value on random probes (away from the boundary)
    rng = np.random.default_rng(5)
    lo, hi = verts.min(axis=0) - 0.5, verts.max(axis=0) + 0.5
    probes = rng.uniform(lo, hi, size=(1500, 2))
    ps, _, _ = o.query(np.c_[probes, np.zeros(len(probes))], nthreads=NT)
    clear = np.abs(ps) > cell
    assert clear.sum() > 800
    assert np.array_equal(_inside(loops, probes[clear]), ps[clear] < 0.0)
    # 4. a finer cell changes the enclosed area by O(cell^2) only
    loops2, st2 = ctx.swept_outline(w["coeffs"], w["T"], cell=cell / 2)
    assert st2["open_chains"] == 0
    a1 = sum(areas)
    a2 = sum(0.5 * np.sum(lp[:, 0] * np.roll(lp[:, 1], -1) - np.roll(lp[:, 0], -1) * lp[:, 1]) for lp in loops2)
    assert abs(a1 - a2) < 0.02 * abs(a2)


def test_swept_outline_argument_errors(built):
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C1", P=50, minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    with pytest.raises(svsdf_amd.SvsdfError):
        ctx.swept_outline(w["coeffs"], w["T"], cell=0.0)
    with pytest.raises(svsdf_amd.SvsdfError):
        ctx.swept_outline(w["coeffs"], w["T"], cell=1e-7)     # more than 1e6 cells per side
    loops, st = ctx.swept_outline(w["coeffs"], w["T"], cell=0.1)   # works without a resident cloud
    assert len(loops) >= 1 and st["open_chains"] == 0
