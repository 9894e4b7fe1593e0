"""BASELINE config 5 on the GPU: "arbitrary .obj mesh ... (no analytic shape SDF)".

Every mesh the reference ships (src/plan_manager/shapes/*.obj; data fixture tests/golden/reference_assets.json) goes
mesh -> z = 0 outline (svsdf_mesh_outline, host C++) -> generic Polygon shape (Polygon::getonlySDF, Shape.hpp:1448-1476)
-> the whole HIP pipeline through the C ABI, against the CPU oracle's plain loop over all edges:
  * reduced cost / gradients at the north-star gates (cost 1e-7, gradient 1e-5) against the oracle of record;
  * per-point SVSDF, t* and gradient direction BIT FOR BIT against the oracle in device-trig mode -- the candidate lists
    (grid cells / slabs) the kernels use instead of the loop over all 77 ... 754 edges change no bit anywhere.
"""
import os

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
NT = os.cpu_count() or 1
MESHES = ["sdArc", "sdCutDisk", "sdHeart", "sdHorseshoe", "sdOrientedVesica", "sdPie", "sdPie2", "sdRhombus",
          "sdRoundedCross", "sdRoundedX", "sdTunnel", "sdUnevenCapsule", "star"]


def _rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


def _mk(name, P, N=8, dist="corridor"):
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(dict(shape="Polygon", N=N, P=P, scenario="star", mesh=name), dist=dist, minco=svsdf_amd.minco_coeffs)
    kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], polygon=w["polygon"],
              head_state=w["head_state"], tail_state=w["tail_state"])
    ctx = svsdf_amd.SvsdfContext(shape="Polygon", device=0, **kw)
    ctx.set_points(w["points"])
    o = orc.Oracle("Polygon", **kw)
    o.set_traj(w["coeffs"], w["T"])
    return w, ctx, o


@pytest.mark.parametrize("name", MESHES)
def test_mesh_outline_shape_matches_oracle(built, name):
    big = name in ("sdArc", "sdRoundedCross")          # 754 / 614 edges: the oracle's loop is what takes the time
    w, ctx, o = _mk(name, 400 if big else 900)
    assert ctx.shape_bound()[1] <= ctx.shape_bound()[0]
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    ocost, ogT, ogC = o.penalty(w["points"], nthreads=NT, sum_mode=1)
    assert ocost > 0
    assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
    assert _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5, (_rel(gC, ogC), _rel(gT, ogT))
    o.set_trig_mode(1)
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    osdf, ots, og = o.query(w["points"], nthreads=NT)
    assert (osdf <= 0).sum() > 20                       # interior points: the GSIP loop ran
    assert np.array_equal(ts, ots) and np.array_equal(sdf, osdf) and np.array_equal(g, og), name


def test_c5_is_the_star_mesh_outline(built):
    """CONFIGS["C5"]: the 77-vertex z = 0 outline of the reference's star.obj, 16 pieces (BASELINE configs[4])."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C5", P=64, minco=svsdf_amd.minco_coeffs)
    assert w["shape"] == "Polygon" and w["polygon"].shape == (77, 2) and len(w["T"]) == 16
    V, F = workload.reference_mesh("star")
    assert V.shape == (152, 3) and F.shape == (300, 3)
    # the outline hugs the analytic star the mesh was made from (r = 2.8): every vertex within 6 cm of its zero set
    sd = orc.Oracle("star").shape_eval(w["polygon"])
    assert np.abs(sd).max() < 0.06


def test_c5_map_distribution_and_far_points(built):
    """Map-uniform cloud (points up to ~70 m from the robot: far outside both candidate grids -> the full-loop path of
    the Polygon evaluation, and the exact cull) against the oracle."""
    w, ctx, o = _mk("star", 3000, N=16, dist="map")
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    st = ctx.stats()
    ocost, ogT, ogC = o.penalty(w["points"], nthreads=NT, sum_mode=1)
    assert st["culled_points"] > 0.3 * 3000
    assert abs(cost - ocost) <= 1e-7 * abs(ocost) and _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5
    o.set_trig_mode(1)
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    osdf, ots, og = o.query(w["points"], nthreads=NT)
    assert np.array_equal(ts, ots) and np.array_equal(sdf, osdf) and np.array_equal(g, og)


def test_c5_negative_gsip_radius_case(built, monkeypatch):
    """C5 at 20 k points holds an interior point whose GSIP radius update r <- r - max_g (sw_manager.hpp:1003-1006) turns
    NEGATIVE (a circle sample's local argmin search ends in a far basin, max_g > r): the reference then samples the circle
    of radius |r| and reports a positive "SVSDF" for an interior point.  k_round's candidate-chunk lists once assumed
    r >= 0 there (round 3): every point, in the full-scan bound mode, must match the oracle bit for bit."""
    monkeypatch.setenv("SVSDF_UB_FULL", "1")
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C5", P=20000, minco=svsdf_amd.minco_coeffs)
    kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], polygon=w["polygon"],
              head_state=w["head_state"], tail_state=w["tail_state"])
    ctx = svsdf_amd.SvsdfContext(shape="Polygon", device=0, **kw)
    ctx.set_points(w["points"])
    o = orc.Oracle("Polygon", **kw)
    o.set_traj(w["coeffs"], w["T"])
    o.set_trig_mode(1)
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    osdf, ots, og = o.query(w["points"], nthreads=NT)
    assert np.array_equal(ts, ots) and np.array_equal(sdf, osdf) and np.array_equal(g, og)
    assert ctx.stats()["gsip_bound_mode"] == 1


def test_large_outline_takes_the_global_memory_path(built):
    """Outlines of more than 1024 vertices do not fit the LDS budget of the solve / round kernels: their edges stay in
    global memory (the plain Polygon kernel variants).  A 1500-gon with a wavy radius against the oracle: gates +
    bit-identity per point; and the same answer when a small outline is forced onto that path (SVSDF_POLY_LDS=0)."""
    import svsdf_amd
    from svsdf_amd import workload
    ang = np.linspace(0, 2 * np.pi, 1500, endpoint=False)
    rad = 2.2 + 0.5 * np.sin(5 * ang) + 0.05 * np.sin(61 * ang)
    poly = np.column_stack([rad * np.cos(ang), rad * np.sin(ang)])
    w = workload.make(dict(shape="Polygon", N=8, P=300, scenario="star"), minco=svsdf_amd.minco_coeffs)
    kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], polygon=poly,
              head_state=w["head_state"], tail_state=w["tail_state"])
    ctx = svsdf_amd.SvsdfContext(shape="Polygon", device=0, **kw)
    ctx.set_points(w["points"])
    o = orc.Oracle("Polygon", **kw)
    o.set_traj(w["coeffs"], w["T"])
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    ocost, ogT, ogC = o.penalty(w["points"], nthreads=NT, sum_mode=1)
    assert abs(cost - ocost) <= 1e-7 * abs(ocost) and _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5
    o.set_trig_mode(1)
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    osdf, ots, og = o.query(w["points"], nthreads=NT)
    assert np.array_equal(ts, ots) and np.array_equal(sdf, osdf) and np.array_equal(g, og)
    # the star outline (77 vertices) through both variants
    ws, c1, _ = _mk("star", 600)
    a = c1.query_points(ws["coeffs"], ws["T"])
    os.environ["SVSDF_POLY_LDS"] = "0"
    try:
        _, c2, _ = _mk("star", 600)
        b = c2.query_points(ws["coeffs"], ws["T"])
    finally:
        del os.environ["SVSDF_POLY_LDS"]
    for u, v in zip(a[:3], b[:3]):
        assert np.array_equal(u, v)


def test_polygon_vertex_limit(built):
    """SVSDF_MAX_POLY_VERTS = 8190 entries of the edge array (round 5; 4096 before): vertices + one closing copy per loop of
    a multi-loop outline."""
    import svsdf_amd
    ang = np.linspace(0, 2 * np.pi, 8190, endpoint=False)
    ok = svsdf_amd.SvsdfContext(shape="Polygon", device=0, polygon=np.column_stack([2 * np.cos(ang), 2 * np.sin(ang)]))
    assert abs(ok.shape_bound()[0] - 2.0) < 1e-3
    ok.close()
    ang = np.linspace(0, 2 * np.pi, 8191, endpoint=False)
    with pytest.raises(svsdf_amd.SvsdfError):
        svsdf_amd.SvsdfContext(shape="Polygon", device=0, polygon=np.column_stack([2 * np.cos(ang), 2 * np.sin(ang)]))
    two = np.concatenate([np.column_stack([2 * np.cos(ang[:8186]), 2 * np.sin(ang[:8186])]), np.array([[5.0, 0], [6, 0], [6, 1], [5, 1]])])
    with pytest.raises(svsdf_amd.SvsdfError):      # 8190 vertices + 2 closing copies
        svsdf_amd.SvsdfContext(shape="Polygon", device=0, polygon=two, polygon_loops=[8186, 4])


def _extrude(loops, z0=-0.5, z1=0.5):
    """A closed triangle mesh whose z = 0 section is exactly the given loops: walls over [z0, z1] (caps are irrelevant to a
    section strictly between them and are left out -- svsdf_mesh_section only looks at straddling triangles)."""
    V, F = [], []
    for lp in loops:
        b = len(V)
        m = len(lp)
        V += [(x, y, z0) for x, y in lp] + [(x, y, z1) for x, y in lp]
        for i in range(m):
            j = (i + 1) % m
            F += [(b + i, b + j, b + m + j), (b + i, b + m + j, b + m + i)]
    return np.array(V, dtype=np.float64), np.array(F, dtype=np.int32)


@pytest.mark.parametrize("case", ["annulus", "two solids"])
def test_multi_loop_mesh_sections_match_oracle(built, case):
    """VERDICT r4 #6 / BASELINE config 5 ("arbitrary .obj mesh"): a mesh whose z = 0 section has a hole (an annulus) or
    consists of two solids.  mesh -> svsdf_mesh_section (both loops) -> Polygon of two loops -> the whole pipeline: gates
    against the oracle of record, per-point SVSDF / t* / gradient bit for bit in device-trig mode -- the oracle being the
    reference's plain loop (Shape.hpp:1448-1476) over the union of the loops' edges."""
    import svsdf_amd
    from svsdf_amd import workload
    a = np.linspace(0, 2 * np.pi, 72, endpoint=False)
    if case == "annulus":      # a ring-shaped robot, 2.4 m across with a 1.3 m hole: obstacles fit inside the hole
        loops = [np.column_stack([2.4 * np.cos(a), 1.7 * np.sin(a)]), np.column_stack([1.3 * np.cos(a[::2]), 0.9 * np.sin(a[::2])])[::-1]]
    else:                      # a body and a detached "sensor mast" 0.6 m off its side
        loops = [np.column_stack([1.9 * np.cos(a), 1.1 * np.sin(a)]), np.column_stack([2.9 + 0.4 * np.cos(a[::4]), 0.3 + 0.4 * np.sin(a[::4])])]
    V, F = _extrude(loops)
    xy, sizes = svsdf_amd.mesh_section(V, F)
    assert sorted(sizes) == sorted(2 * len(lp) for lp in loops) and len(sizes) == 2   # (every wall quad's diagonal adds a crossing point)
    w = workload.make(dict(shape="Polygon", N=8, P=1500, scenario="star"), minco=svsdf_amd.minco_coeffs)
    kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], polygon=xy, polygon_loops=sizes,
              head_state=w["head_state"], tail_state=w["tail_state"])
    ctx = svsdf_amd.SvsdfContext(shape="Polygon", device=0, **kw)
    ctx.set_points(w["points"])
    o = orc.Oracle("Polygon", **kw)
    o.set_traj(w["coeffs"], w["T"])
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    ocost, ogT, ogC = o.penalty(w["points"], nthreads=NT, sum_mode=1)
    assert ocost > 0
    assert abs(cost - ocost) <= 1e-7 * abs(ocost), (cost, ocost)
    assert _rel(gC, ogC) <= 1e-5 and _rel(gT, ogT) <= 1e-5, (_rel(gC, ogC), _rel(gT, ogT))
    o.set_trig_mode(1)
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    osdf, ots, og = o.query(w["points"], nthreads=NT)
    assert (osdf <= 0).sum() > 20
    assert np.array_equal(ts, ots) and np.array_equal(sdf, osdf) and np.array_equal(g, og), case
    # not the single chain over the same vertices (its bridge edge would be part of the robot)
    o1 = orc.Oracle("Polygon", **{**kw, "polygon_loops": None})
    o1.set_traj(w["coeffs"], w["T"])
    o1.set_trig_mode(1)
    assert not np.array_equal(o1.query(w["points"], nthreads=NT)[0], osdf)
    # the swept-volume outline runs a private context with the same shape: it must carry the loops too
    out = ctx.swept_outline(w["coeffs"], w["T"], cell=0.25)
    assert len(out[0]) >= 1 and all(len(lp) >= 3 for lp in out[0])
