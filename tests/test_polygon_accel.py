"""BASELINE config 5 on the CPU: mesh (.obj) -> z = 0 outline -> generic Polygon shape.

* svsdf_mesh_outline (product, host C++) on the 13 meshes the reference ships (src/plan_manager/shapes/*.obj, committed
  as data in tests/golden/reference_assets.json): one closed loop each, identical to an independent numpy slicing, and
  consistent with the mesh itself -- the exact generalised winding number of the 3-D mesh (sum of signed solid angles,
  FP64, brute force over the triangles: what igl::fast_winding_number approximates, Shape.hpp:332-340) decides
  inside / outside at z = 0 exactly like the crossing parity of the outline, and the 3-D distance to the mesh equals the
  2-D distance to the outline up to the meshes' own wall slant.
* the product's Polygon evaluation (csrc/svsdf_polygon.hpp: candidate lists per grid cell / slab instead of the
  reference's loop over all edges) compiled for the HOST (tests/cpp/poly_host.cpp, the same __host__ __device__
  functions the gfx950 kernels inline) is bit-identical -- value, closest point, analytic gradient -- to the oracle's
  plain loop (Polygon::getonlySDF / getonlyGrad1, Shape.hpp:1448-1531) on all 13 outlines, including queries on
  vertices, on edges, on the horizontal lines through vertices, and far outside the grids.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "cpp", "libpoly_host.so")
EXPECTED_VERTS = {"sdArc": 754, "sdCutDisk": 98, "sdHeart": 101, "sdHorseshoe": 138, "sdOrientedVesica": 104,
                  "sdPie": 138, "sdPie2": 91, "sdRhombus": 80, "sdRoundedCross": 614, "sdRoundedX": 96,
                  "sdTunnel": 96, "sdUnevenCapsule": 111, "star": 77}


def _names():
    from svsdf_amd import workload
    return workload.MESH_NAMES


@pytest.fixture(scope="module")
def polyhost(built):
    src = os.path.join(ROOT, "tests", "cpp", "poly_host.cpp")
    hdr = os.path.join(ROOT, "implicit-svsdf-planner_amd", "csrc", "svsdf_polygon.hpp")
    if not os.path.exists(SO) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(SO):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc if os.path.exists(hipcc) else "hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-mfma",
                               "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", src, "-o", SO])
    return C.CDLL(SO)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _slice_numpy(V, F, z0=0.0):
    """Independent restatement of the cross-section: crossing points per straddling triangle, chained."""
    up = V[:, 2] >= z0
    pts, adj = {}, {}
    for f in F:
        ks = []
        for i in range(3):
            a, b = int(f[i]), int(f[(i + 1) % 3])
            if up[a] != up[b]:
                k = (min(a, b), max(a, b))
                if k not in pts:
                    t = (z0 - V[k[0], 2]) / (V[k[1], 2] - V[k[0], 2])
                    pts[k] = V[k[0], :2] + t * (V[k[1], :2] - V[k[0], :2])
                ks.append(k)
        if ks:
            assert len(ks) == 2
            adj.setdefault(ks[0], []).append(ks[1])
            adj.setdefault(ks[1], []).append(ks[0])
    assert all(len(v) == 2 for v in adj.values())   # closed manifold section
    return pts, adj


def _winding_number(V, F, q):
    """Exact generalised winding number of the closed triangle mesh at the points q (m, 3): sum of the signed solid
    angles of the triangles (van Oosterom-Strackee) / 4 pi."""
    a = V[F[:, 0]][None] - q[:, None]
    b = V[F[:, 1]][None] - q[:, None]
    c = V[F[:, 2]][None] - q[:, None]
    la, lb, lc = (np.linalg.norm(v, axis=-1) for v in (a, b, c))
    num = np.einsum("mfi,mfi->mf", a, np.cross(b, c))
    den = la * lb * lc + np.einsum("mfi,mfi->mf", a, b) * lc + np.einsum("mfi,mfi->mf", b, c) * la + np.einsum("mfi,mfi->mf", c, a) * lb
    return (2.0 * np.arctan2(num, den)).sum(axis=1) / (4.0 * np.pi)


def _mesh_distance(V, F, q):
    """Brute-force unsigned distance from q (m, 3) to the triangles (closest point via barycentric clamping, Ericson)."""
    out = np.full(len(q), np.inf)
    for f in F:
        a, b, c = V[f[0]], V[f[1]], V[f[2]]
        ab, ac = b - a, c - a
        ap = q - a
        d1, d2 = ap @ ab, ap @ ac
        bp = q - b
        d3, d4 = bp @ ab, bp @ ac
        cp = q - c
        d5, d6 = cp @ ab, cp @ ac
        vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
        denom = va + vb + vc
        denom = np.where(denom == 0.0, 1.0, denom)
        v, w = vb / denom, vc / denom
        cl = a + np.outer(v, ab) + np.outer(w, ac)                                   # interior
        def put(mask, pt):
            cl[mask] = pt[mask] if pt.ndim == 2 else pt
        e_ab = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
        put(e_ab, a + np.outer(d1 / np.where(d1 - d3 == 0, 1.0, d1 - d3), ab))
        e_ac = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
        put(e_ac, a + np.outer(d2 / np.where(d2 - d6 == 0, 1.0, d2 - d6), ac))
        e_bc = (va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0)
        tt = (d4 - d3) / np.where((d4 - d3) + (d5 - d6) == 0, 1.0, (d4 - d3) + (d5 - d6))
        put(e_bc, b + np.outer(tt, c - b))
        put((d1 <= 0) & (d2 <= 0), a)
        put((d3 >= 0) & (d4 <= d3), b)
        put((d6 >= 0) & (d5 <= d6), c)
        out = np.minimum(out, np.linalg.norm(q - cl, axis=1))
    return out


def test_mesh_outline_of_every_reference_mesh(built):
    import svsdf_amd
    from svsdf_amd import workload
    for name in _names():
        V, F = workload.reference_mesh(name)
        xy, loops = svsdf_amd.mesh_outline(V, F, 0.0)
        assert loops == 1 and len(xy) == EXPECTED_VERTS[name], (name, loops, len(xy))
        pts, adj = _slice_numpy(V, F)
        assert len(pts) == len(xy)
        # same point set, and consecutive outline vertices are neighbours in the independent chaining
        key_of = {tuple(np.round(p, 12)): k for k, p in pts.items()}
        ks = [key_of[tuple(np.round(p, 12))] for p in xy]
        assert len(set(ks)) == len(ks)
        for i, k in enumerate(ks):
            assert ks[(i + 1) % len(ks)] in adj[k], name
        # no consecutive duplicates, finite, inside the mesh's xy bounding box
        assert np.isfinite(xy).all() and (np.linalg.norm(xy - np.roll(xy, -1, 0), axis=1) > 0).all()
        assert (xy.min(0) >= V[:, :2].min(0) - 1e-12).all() and (xy.max(0) <= V[:, :2].max(0) + 1e-12).all()


def test_obj_reader_and_errors(built, tmp_path):
    import svsdf_amd
    from svsdf_amd import workload
    V, F = workload.reference_mesh("star")
    p = tmp_path / "m.obj"
    with open(p, "w") as f:
        f.write("# comment\no thing\n")
        for v in V:
            f.write("v %.6f %.6f %.6f\n" % tuple(v))
        f.write("vn 0 0 1\n")
        for k, t in enumerate(F):
            if k % 3 == 0:
                f.write("f %d %d %d\n" % tuple(t + 1))
            elif k % 3 == 1:
                f.write("f %d/1/1 %d/2/1 %d//1\n" % tuple(t + 1))
            else:
                f.write("f %d %d %d\n" % tuple(t - len(V)))     # negative (relative) indices
    a, la = svsdf_amd.mesh_outline_obj(p)
    b, lb = svsdf_amd.mesh_outline(V, F)
    assert la == lb == 1 and np.array_equal(a, b)
    # a quad face is fanned: a unit cube made of 6 quads, sliced at z = 0.25 -> its square
    q = tmp_path / "cube.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\n"
                 "f 1 2 3 4\nf 5 8 7 6\nf 1 5 6 2\nf 2 6 7 3\nf 3 7 8 4\nf 4 8 5 1\n")
    sq, loops = svsdf_amd.mesh_outline_obj(q, 0.25)
    assert loops == 1 and len(sq) == 8    # 4 corners + 4 points on the face diagonals
    assert {(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)} <= set(map(tuple, np.round(sq, 12)))
    area = 0.5 * abs(np.dot(sq[:, 0], np.roll(sq[:, 1], -1)) - np.dot(sq[:, 1], np.roll(sq[:, 0], -1)))
    assert abs(area - 1.0) < 1e-12
    # two disjoint solids: two closed loops, the longer one is returned
    V2 = np.concatenate([V, V[:8] * 0.0 + np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]]) + np.array([20.0, 0.0, -0.5])])
    quads = np.array([[0, 1, 2, 3], [4, 7, 6, 5], [0, 4, 5, 1], [1, 5, 6, 2], [2, 6, 7, 3], [3, 7, 4, 0]]) + len(V)
    F2 = np.concatenate([F, np.concatenate([quads[:, [0, 1, 2]], quads[:, [0, 2, 3]]])]).astype(np.int32)
    both, loops2 = svsdf_amd.mesh_outline(V2, F2)
    assert loops2 == 2 and np.array_equal(both, b)
    # the main loop is the one enclosing the largest AREA, not the one with the most vertices: a coarse 10 x 10 box
    # (8 crossing points) next to the finely tessellated star (77)
    V3 = V2.copy()
    V3[len(V):, :2] = (V3[len(V):, :2] - np.array([20.0, 0.0])) * 10.0 + np.array([20.0, 0.0])
    big, loops3 = svsdf_amd.mesh_outline(V3, F2)
    assert loops3 == 2 and len(big) == 8 and big[:, 0].min() >= 20.0 - 1e-12
    # ... and the whole section (round 5: the mirrors plan with every loop -- rounds 3-4 refused such meshes): both loops,
    # the one enclosing the larger area first, each the loop mesh_outline gives for its body alone
    two = tmp_path / "two_bodies.obj"
    with open(two, "w") as f:
        for v in V3:
            f.write("v %.9f %.9f %.9f\n" % tuple(v))
        for t in F2:
            f.write("f %d %d %d\n" % tuple(t + 1))
    sec, sizes = svsdf_amd.mesh_section(V3, F2)
    assert sizes == [8, 77] and np.array_equal(sec[:8], big) and np.array_equal(sec[8:], b)
    sec_f, sizes_f = svsdf_amd.mesh_section_obj(two)
    assert sizes_f == [8, 77] and np.abs(sec_f - sec).max() < 1e-8    # (%.9f in the file)
    one, size1 = svsdf_amd.mesh_section(V, F)
    assert size1 == [77] and np.array_equal(one, b)
    with pytest.raises(svsdf_amd.SvsdfError):
        svsdf_amd.mesh_outline_obj(tmp_path / "missing.obj")
    with pytest.raises(svsdf_amd.SvsdfError):
        svsdf_amd.mesh_outline(V, F, 5.0)       # the plane misses the mesh
    with pytest.raises(svsdf_amd.SvsdfError):
        svsdf_amd.mesh_outline(V, np.array([[0, 1, 10 ** 6]], dtype=np.int32))   # index out of range


@pytest.mark.parametrize("name", ["star", "sdHorseshoe", "sdHeart", "sdArc", "sdCutDisk", "sdTunnel"])
def test_outline_agrees_with_the_mesh_itself(built, name):
    """Oracle of config 5 as SURVEY.md §8(c) defines it: exact generalised winding number x brute-force distance of the
    3-D mesh, against Polygon::getonlySDF of the outline, at z = 0."""
    from svsdf_amd import workload
    V, F = workload.reference_mesh(name)
    xy = workload.mesh_outline(name)
    rng = np.random.default_rng(3)
    lo, hi = xy.min(0) - 1.5, xy.max(0) + 1.5
    q2 = rng.uniform(lo, hi, (1500, 2))
    q3 = np.concatenate([q2, np.zeros((len(q2), 1))], axis=1)
    sdf = orc.Oracle("Polygon", polygon=xy).shape_eval(q2)
    wn = np.abs(_winding_number(V, F, q3))
    clear = np.abs(sdf) > 1e-6
    assert (np.abs(wn[clear] - np.round(wn[clear])) < 1e-6).all()            # a closed mesh: integer winding numbers
    assert ((wn[clear] > 0.5) == (sdf[clear] < 0)).all(), name                # inside <=> odd crossing parity
    # distance: the outline lies ON the mesh, so the mesh is never farther than the outline; the slabs' walls are
    # vertical up to the meshes' own slant, so outside it is not much nearer either.  Inside, the caps (|z| ~ 0.5)
    # take over once the outline is farther than they are.
    d3 = _mesh_distance(V, F, q3)
    assert (d3 <= np.abs(sdf) + 1e-9).all()
    out = sdf > 0
    assert np.abs(d3[out] - sdf[out]).max() < 0.03, (name, np.abs(d3[out] - sdf[out]).max())
    zcap = np.abs(V[:, 2]).max()
    inn = sdf < 0
    assert (d3[inn] >= np.minimum(-sdf[inn], zcap) - 0.06).all()


def test_polygon_candidate_lists_are_bit_identical_to_the_plain_loop(polyhost):
    from svsdf_amd import workload
    rng = np.random.default_rng(5)
    for name in _names():
        xy = np.ascontiguousarray(workload.mesh_outline(name))
        n = len(xy)
        size = max(np.ptp(xy[:, 0]), np.ptp(xy[:, 1]))
        c = 0.5 * (xy.min(0) + xy.max(0))
        pts = np.ascontiguousarray(np.concatenate([
            c + rng.uniform(-0.7 * size, 0.7 * size, (12000, 2)),       # fine grid
            c + rng.uniform(-3.4 * size, 3.4 * size, (12000, 2)),       # coarse grid
            c + rng.uniform(-9.0 * size, 9.0 * size, (6000, 2)),        # far grid
            c + rng.uniform(-40.4 * size, 40.4 * size, (6000, 2)),      # far grid and its rim
            c + rng.uniform(-90.0 * size, 90.0 * size, (2000, 2)),      # beyond all three: the lane's own loop over all edges
            c + (rng.uniform(-0.75 * size, 0.75 * size, (6000, 2)) // (1.5 * size / 128)) * (1.5 * size / 128),   # on fine-grid lines
            xy, 0.5 * (xy + np.roll(xy, -1, 0)),                        # on vertices / on edges
            np.stack([c[0] + rng.uniform(-2 * size, 2 * size, n), xy[:, 1]], 1),   # rays through vertices
            np.stack([xy[:, 0], c[1] + rng.uniform(-2 * size, 2 * size, n)], 1),
            xy + rng.normal(0, 1e-12, xy.shape), xy + rng.normal(0, 1e-6, xy.shape), xy + rng.normal(0, 1e-2, xy.shape)]))
        o = orc.Oracle("Polygon", polygon=xy)
        so, go = o.shape_eval(pts, grad=True)
        s, sc, cl = np.zeros(len(pts)), np.zeros(len(pts)), np.zeros((len(pts), 2))
        st = np.zeros(6, dtype=np.int64)
        assert polyhost.polyhost_eval(_dp(xy), n, _dp(pts), C.c_size_t(len(pts)), _dp(s), _dp(sc), _dp(cl),
                                      st.ctypes.data_as(C.POINTER(C.c_longlong))) == 0
        lv, cnt = np.zeros(len(pts), dtype=np.int32), np.zeros(len(pts), dtype=np.int32)
        ip = C.POINTER(C.c_int)
        polyhost.polyhost_visits(_dp(xy), n, _dp(pts), C.c_size_t(len(pts)), lv.ctypes.data_as(ip), cnt.ctypes.data_as(ip))
        assert {0, 1, 2, 3} <= set(lv.tolist())                   # all four paths exercised
        # the analytic gradient the device derives from the closest point (shape_grad, svsdf_shapes.hpp)
        v = pts - cl
        z = (v * v).sum(1)
        g = np.where(z[:, None] > 0, v / np.sqrt(np.where(z > 0, z, 1.0))[:, None], v)
        g = np.where(np.signbit(sc)[:, None], -g, g)
        i64 = lambda a: np.ascontiguousarray(a).view(np.int64)
        assert (i64(s) == i64(so)).all(), name
        assert (i64(sc) == i64(so)).all(), name
        assert (i64(g) == i64(go)).all(), name
        # the point of the exercise: far fewer edges than the reference's loop over all of them
        assert cnt[lv == 0].mean() < max(4.0, 0.03 * n) and cnt[lv == 1].mean() < max(6.0, 0.04 * n) and cnt[lv == 2].mean() < max(6.0, 0.02 * n), \
            (name, cnt[lv == 0].mean(), cnt[lv == 1].mean(), cnt[lv == 2].mean())


def test_polygon_degenerate_outlines(polyhost):
    """Repeated vertices (zero-length edges: 0/0 in the reference's dis2Seg, never the minimum), collinear runs, a thin
    sliver, and the reference's own fallback rectangle (SWM:363-369)."""
    rng = np.random.default_rng(9)
    outlines = [
        np.array([[6, -0.1], [6, 0.1], [-6, 0.1], [-6, -0.1]], dtype=float),
        np.array([[0, 0], [1, 0], [1, 0], [2, 0], [2, 1], [2, 1], [0, 1]], dtype=float),
        np.array([[0, 0], [1, 0], [2, 0], [3, 0], [3, 2], [1.5, 2], [0, 2], [0, 1]], dtype=float),
        np.array([[0, 0], [10, 1e-7], [0, 2e-7]], dtype=float),
    ]
    for xy in outlines:
        xy = np.ascontiguousarray(xy)
        size = max(np.ptp(xy[:, 0]), np.ptp(xy[:, 1]))
        c = 0.5 * (xy.min(0) + xy.max(0))
        pts = np.ascontiguousarray(np.concatenate([c + rng.uniform(-1.2 * size, 1.2 * size, (6000, 2)),
                                                   c + rng.uniform(-5 * size, 5 * size, (3000, 2)), xy,
                                                   np.stack([c[0] + rng.uniform(-size, size, len(xy)), xy[:, 1]], 1)]))
        so = orc.Oracle("Polygon", polygon=xy).shape_eval(pts)
        s = np.zeros(len(pts))
        assert polyhost.polyhost_eval(_dp(xy), len(xy), _dp(pts), C.c_size_t(len(pts)), _dp(s), None, None, None) == 0
        assert (s.view(np.int64) == so.view(np.int64)).all()
    # refused: too few / too many vertices, non-finite
    s = np.zeros(1)
    p = np.zeros((1, 2))
    assert polyhost.polyhost_eval(_dp(np.zeros((2, 2))), 2, _dp(p), C.c_size_t(1), _dp(s), None, None, None) == 1
    bad = np.array([[0, 0], [1, np.nan], [1, 1]])
    assert polyhost.polyhost_eval(_dp(bad), 3, _dp(p), C.c_size_t(1), _dp(s), None, None, None) == 1


def test_poly_quot_is_the_division(polyhost):
    """The quotient of dis2Seg from the edge's reciprocal (poly_quot: two residual steps) against a / b itself, on
    operands built to sit on rounding boundaries: b with a significand of all ones / near a power of two, a = b * t
    moved by an ulp either way, tiny and huge |a|, zeros, NaN."""
    rng = np.random.default_rng(9)
    m = 2000000
    def rnd(mode):
        man = rng.integers(0, 1 << 52, m, dtype=np.uint64)
        if mode == 1:
            man |= np.uint64(0xFFFFFFFFFF000)
        elif mode == 2:
            man &= np.uint64(0xFFF)
        elif mode == 3:
            man = np.uint64(0xFFFFFFFFFFFFF) - (man & np.uint64(7))
        return man
    b = ((np.uint64(1023 - 40) + rng.integers(0, 60, m).astype(np.uint64)) << np.uint64(52) | rnd(rng.integers(0, 4))).view(np.float64)
    a = b * rng.uniform(-0.5, 1.5, m)
    a = (a.view(np.int64) + rng.integers(-1, 2, m)).view(np.float64)
    a[:1000] = [0.0, -0.0, 1e-160, -1e-160, 1e160, np.nan, np.inf, 5e-324, 1e-151, 1e151] * 100
    b[1000:1100] = 0.0                                                   # zero-length edge: 0 / 0
    a[1000:1100] = 0.0
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    q = np.zeros(m)
    used = np.zeros(m, dtype=np.int32)
    polyhost.polyhost_quot(_dp(a), _dp(b), C.c_size_t(m), _dp(q), used.ctypes.data_as(C.POINTER(C.c_int)))
    with np.errstate(all="ignore"):
        ref = a / b
    assert (q.view(np.int64) == ref.view(np.int64)).all()
    assert used.mean() > 0.99 and not used[:8].any()                       # the refinement served all but the odd ones


def test_cell_lookup_agrees_with_the_cells_the_lists_were_built_for(polyhost):
    """poly_locate computes a cell index from the distance to the levels' common centre; the host lays the cells out
    from each level's corner.  The two may differ by rounding only: every query must lie within 1e-9 sizes of the
    rectangle of the cell it is given (the lists hold for the rectangle enlarged by 1e-7 sizes), on grid lines too."""
    from svsdf_amd import workload
    rng = np.random.default_rng(21)
    for name in ("star", "sdHeart", "sdArc"):
        xy = np.ascontiguousarray(workload.mesh_outline(name) + np.array([37.25, -11.5]))   # off-centre on purpose
        n = len(xy)
        size = max(np.ptp(xy[:, 0]), np.ptp(xy[:, 1]))
        c = 0.5 * (xy.min(0) + xy.max(0))
        ext = np.array([1.5, 7.0, 81.0]) * size
        pts = [c + rng.uniform(-0.5 * e, 0.5 * e, (20000, 2)) for e in ext]
        for e in ext:                                    # on and next to the grid lines of every level
            k = rng.integers(-128, 129, (20000, 2))
            q = c + k * (e / (128 if e == ext[0] else 256)) * (1.0 if e != ext[0] else 1.0)
            pts += [q, np.nextafter(q, np.inf), np.nextafter(q, -np.inf)]
        pts = np.ascontiguousarray(np.concatenate(pts))
        rect = np.zeros((len(pts), 4))
        lv = np.zeros(len(pts), dtype=np.int32)
        assert polyhost.polyhost_cell_rect(_dp(xy), n, _dp(pts), C.c_size_t(len(pts)), _dp(rect),
                                           lv.ctypes.data_as(C.POINTER(C.c_int))) == 0
        inside = lv < 3
        assert inside.mean() > 0.9 and {0, 1, 2} <= set(lv.tolist())
        tol = 1e-9 * size
        p, r = pts[inside], rect[inside]
        assert (p[:, 0] >= r[:, 0] - tol).all() and (p[:, 0] <= r[:, 2] + tol).all()
        assert (p[:, 1] >= r[:, 1] - tol).all() and (p[:, 1] <= r[:, 3] + tol).all()
        # the innermost level that holds the query is used (a query well inside the fine extent gets a fine cell)
        m = np.max(np.abs(pts - c), axis=1)
        assert (lv[m < 0.74 * size] == 0).all() and (lv[(m > 0.76 * size) & (m < 3.49 * size)] == 1).all()
        assert (lv[(m > 3.51 * size) & (m < 40.4 * size)] == 2).all() and (lv[m > 40.6 * size] == 3).all()


def test_polygon_outlines_outside_the_quotient_range(polyhost):
    """Outlines so small / large that v.v leaves [1e-100, 1e100]: PolyAccel::div_ok is off and every lane evaluates on its
    own path (the division itself); and an outline far from the origin.  Bit-identical to the oracle's plain loop."""
    from svsdf_amd import workload
    rng = np.random.default_rng(33)
    base = np.ascontiguousarray(workload.mesh_outline("star"))
    for scale, shift in ((1e-60, 0.0), (1e60, 0.0), (1.0, 1e6), (1e-3, -250.0)):
        xy = np.ascontiguousarray(base * scale + shift)
        size = max(np.ptp(xy[:, 0]), np.ptp(xy[:, 1]))
        c = 0.5 * (xy.min(0) + xy.max(0))
        pts = np.ascontiguousarray(np.concatenate([c + rng.uniform(-0.7 * size, 0.7 * size, (8000, 2)),
                                                   c + rng.uniform(-6 * size, 6 * size, (4000, 2)),
                                                   c + rng.uniform(-60 * size, 60 * size, (2000, 2)), xy]))
        so, go = orc.Oracle("Polygon", polygon=xy).shape_eval(pts, grad=True)
        s, sc, cl = np.zeros(len(pts)), np.zeros(len(pts)), np.zeros((len(pts), 2))
        assert polyhost.polyhost_eval(_dp(xy), len(xy), _dp(pts), C.c_size_t(len(pts)), _dp(s), _dp(sc), _dp(cl), None) == 0
        assert (s.view(np.int64) == so.view(np.int64)).all(), (scale, shift)
        assert (sc.view(np.int64) == so.view(np.int64)).all(), (scale, shift)


def _ring(r, n, phase=0.0, centre=(0.0, 0.0), clockwise=False):
    a = phase + 2 * np.pi * np.arange(n) / n
    if clockwise:
        a = a[::-1]
    return np.stack([centre[0] + r * np.cos(a), centre[1] + r * np.sin(a)], 1)


def _multi_loop_outlines():
    """(name, xy, loop sizes): an annulus (a hole), two separate solids, a solid with two holes next to a second solid
    (loop orientation is irrelevant to Polygon::getonlySDF: a minimum and a crossing parity), and the section of the
    two-body mesh of test_obj_reader_and_errors."""
    star = np.array([[np.cos(a) * (1.0 if k % 2 else 2.2), np.sin(a) * (1.0 if k % 2 else 2.2)] for k, a in enumerate(np.linspace(0, 2 * np.pi, 10, endpoint=False))])
    return [
        ("annulus", np.concatenate([_ring(2.0, 60), _ring(1.2, 40, 0.3, clockwise=True)]), [60, 40]),
        ("two solids", np.concatenate([star, _ring(0.8, 12, 0.1, (5.5, 0.7))]), [10, 12]),
        ("two holes + a second solid", np.concatenate([_ring(3.0, 90), _ring(0.6, 20, 0.0, (-1.3, 0.2)), _ring(0.9, 25, 0.5, (1.2, -0.4)), _ring(0.7, 16, 0.2, (4.6, 3.0))]), [90, 20, 25, 16]),
    ]


def test_multi_loop_outlines_are_bit_identical_to_the_plain_loop(polyhost):
    """a section with a hole or of two solids.  The reference's Polygon::getonlySDF (Shape.hpp:1448-1476) is
    a minimum over edges and a crossing count over edges; the oracle evaluates the union of the loops' edges with its plain
    loop (orc_shape_set_loops), the product from its candidate lists with every loop closed by a dead copy of its first
    vertex (svsdf_polygon.hpp): value, closest point and analytic gradient must agree in every bit -- inside the hole,
    between the solids, on vertices and edges of every loop, far away."""
    rng = np.random.default_rng(17)
    ip = C.POINTER(C.c_int)
    for name, xy, sizes in _multi_loop_outlines():
        xy = np.ascontiguousarray(xy)
        n = len(xy)
        size = max(np.ptp(xy[:, 0]), np.ptp(xy[:, 1]))
        c = 0.5 * (xy.min(0) + xy.max(0))
        pts = np.ascontiguousarray(np.concatenate([
            c + rng.uniform(-0.7 * size, 0.7 * size, (12000, 2)), c + rng.uniform(-3.4 * size, 3.4 * size, (8000, 2)),
            c + rng.uniform(-30 * size, 30 * size, (4000, 2)), c + rng.uniform(-90 * size, 90 * size, (1000, 2)),
            xy, 0.5 * (xy + np.roll(xy, -1, 0)),
            np.stack([c[0] + rng.uniform(-2 * size, 2 * size, n), xy[:, 1]], 1),       # rays through vertices of every loop
            xy + rng.normal(0, 1e-12, xy.shape), xy + rng.normal(0, 1e-3, xy.shape)]))
        o = orc.Oracle("Polygon", polygon=xy, polygon_loops=sizes)
        so, go = o.shape_eval(pts, grad=True)
        ls = (C.c_int * len(sizes))(*sizes)
        polyhost.polyhost_set_loops(ls, len(sizes))
        try:
            s, sc, cl = np.zeros(len(pts)), np.zeros(len(pts)), np.zeros((len(pts), 2))
            assert polyhost.polyhost_eval(_dp(xy), n, _dp(pts), C.c_size_t(len(pts)), _dp(s), _dp(sc), _dp(cl), None) == 0
        finally:
            polyhost.polyhost_set_loops(ls, 0)
        v = pts - cl
        z = (v * v).sum(1)
        g = np.where(z[:, None] > 0, v / np.sqrt(np.where(z > 0, z, 1.0))[:, None], v)
        g = np.where(np.signbit(sc)[:, None], -g, g)
        i64 = lambda a: np.ascontiguousarray(a).view(np.int64)
        assert (i64(s) == i64(so)).all(), name
        assert (i64(sc) == i64(so)).all(), name
        assert (i64(g) == i64(go)).all(), name
        # the loops mean what they should: the hole of the annulus is outside, the ring inside
        if name == "annulus":
            probe = np.array([[0.0, 0.0], [1.6, 0.0], [2.5, 0.0]])
            sp = o.shape_eval(probe)
            assert sp[0] > 0 and sp[1] < 0 and sp[2] > 0 and abs(sp[0] - 1.2 * np.cos(np.pi / 40)) < 1e-9
        # a single chain over the same vertices is a DIFFERENT polygon (its bridge edges): the loop table matters
        assert (orc.Oracle("Polygon", polygon=xy).shape_eval(pts) != so).any()
    # loop sizes that do not add up are refused by both
    bad = (C.c_int * 2)(3, 4)
    polyhost.polyhost_set_loops(bad, 2)
    s1 = np.zeros(1)
    assert polyhost.polyhost_eval(_dp(np.zeros((6, 2))), 6, _dp(np.zeros((1, 2))), C.c_size_t(1), _dp(s1), None, None, None) == 1
    polyhost.polyhost_set_loops(bad, 0)
    with pytest.raises(ValueError):
        orc.Oracle("Polygon", polygon=np.zeros((6, 2)), polygon_loops=[3, 4])


def test_candidate_lists_are_built_fast(polyhost):
    """svsdf_create spent 452 / 669 ms building the lists of the 614 / 754-vertex outlines (the
    plain pass kept ~ 100 first-pass candidates per cell before refining them to ~ 5).  The pyramid descent never holds
    more than a parent's short list; bound here with a wide margin for a loaded CI host."""
    from svsdf_amd import workload
    for name in ("sdArc", "sdRoundedCross", "star"):
        xy = np.ascontiguousarray(workload.mesh_outline(name))
        h, ms = C.c_ulonglong(), C.c_double()
        best = 1e9
        for _ in range(3):
            assert polyhost.polyhost_build_hash(_dp(xy), len(xy), C.byref(h), C.byref(ms), None) == 0
            best = min(best, ms.value)
        print(f"{name}: {len(xy)} vertices, lists built in {best:.0f} ms")
        assert best < 400.0, (name, best)
