"""Host code of the swept-volume outline (csrc/svsdf_contour.hpp; SURVEY §8 f4 second half): hierarchical narrow band +
marching squares, compiled for the host (tests/cpp/contour_host.cpp) and run on analytic fields."""
import math
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("contour") / "contour_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "cpp", "contour_host.cpp")])
    return exe


def _run(exe, field, h, levels):
    out = subprocess.check_output([exe, field, repr(h), str(levels)]).decode().split("\n")
    nodes, dense, open_chains, nloops = (int(v) for v in out[0].split())
    loops = [(int(a), float(b)) for a, b in (line.split() for line in out[1:1 + nloops])]
    sx, sy, maxres = (float(v) for v in out[1 + nloops].split())
    bad, vol = out[2 + nloops].split()
    return dict(nodes=nodes, dense=dense, open=open_chains, loops=loops, sx=sx, sy=sy, maxres=maxres,
                bad_edges=int(bad), volume=float(vol))


def test_disc_area_orientation_and_convergence(harness):
    exact = math.pi * 1.7 ** 2
    errs = []
    for h in (0.1, 0.05, 0.025):
        r = _run(harness, "disc", h, 4)
        assert r["open"] == 0 and len(r["loops"]) == 1
        n, area = r["loops"][0]
        assert area > 0.0                      # inside on the left: an outer boundary runs counter-clockwise
        errs.append(abs(area - exact))
        assert r["maxres"] < 0.5 * h * h       # vertices lie on the zero set up to the interpolation error
    assert errs[0] > 3.0 * errs[1] > 9.0 * errs[2] > 0.0   # second order in the cell size


def test_two_components_and_a_hole(harness):
    r = _run(harness, "two", 0.05, 4)
    assert r["open"] == 0 and len(r["loops"]) == 2
    areas = sorted(a for _, a in r["loops"])
    assert areas[0] == pytest.approx(math.pi * 0.64, rel=2e-3) and areas[1] == pytest.approx(math.pi, rel=2e-3)
    r = _run(harness, "ring", 0.05, 4)
    assert r["open"] == 0 and len(r["loops"]) == 2
    areas = sorted(a for _, a in r["loops"])
    assert areas[0] == pytest.approx(-math.pi * 1.5 ** 2, rel=2e-3)   # the hole runs clockwise
    assert areas[1] == pytest.approx(math.pi * 2.5 ** 2, rel=2e-3)


@pytest.mark.parametrize("field", ["disc", "two", "ring", "saddle", "steep"])
def test_band_gives_the_dense_result_with_a_fraction_of_the_nodes(harness, field):
    band = _run(harness, field, 0.05, 4)
    dense = _run(harness, field, 0.05, 0)
    assert dense["nodes"] == dense["dense"]
    assert band["nodes"] < 0.15 * dense["nodes"]
    assert band["open"] == 0 and dense["open"] == 0      # "steep" is not 1-Lipschitz and jumps: the band alone would miss cells
    # the extrusion of marching-squares loops (runs of collinear vertices, holes) closes: walls + caps, volume = area x 1
    assert band["bad_edges"] == 0
    assert band["volume"] == pytest.approx(sum(a for _, a in band["loops"]), rel=1e-9)
    # same cells marched where the zero set is -> the very same vertices (crossing points depend on two node values only)
    assert sorted(band["loops"]) == sorted(dense["loops"])
    assert band["sx"] == pytest.approx(dense["sx"], abs=1e-9) and band["sy"] == pytest.approx(dense["sy"], abs=1e-9)


def _closed_surface_checks(V, F, area, height):
    import numpy as np
    # every directed edge has its reverse exactly once: closed, consistently oriented 2-manifold
    edges = {}
    for t in F:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            edges[(int(a), int(b))] = edges.get((int(a), int(b)), 0) + 1
    assert all(c == 1 for c in edges.values())
    assert all((b, a) in edges for (a, b) in edges)
    # volume by the divergence theorem = area x height, outward orientation
    tri = V[F]
    vol = np.einsum("ij,ij->i", tri[:, 0], np.cross(tri[:, 1], tri[:, 2])).sum() / 6.0
    assert vol == __import__("pytest").approx(area * height, rel=1e-9)


def test_outline_extrusion_walls_and_caps(built):
    """svsdf_outline_extrude (host only): a counter-clockwise square with a clockwise square hole."""
    import numpy as np
    import svsdf_amd
    outer = np.array([[0, 0], [4, 0], [4, 4], [0, 4]], dtype=float)            # inside on the left
    hole = np.array([[1, 1], [1, 3], [3, 3], [3, 1]], dtype=float)             # clockwise
    V, F = svsdf_amd.outline_extrude([outer, hole], z0=-0.5, z1=0.5, caps=False)
    assert V.shape == (16, 3) and F.shape == (16, 3)
    assert set(np.unique(V[:, 2])) == {-0.5, 0.5}
    tri = V[F]
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert np.allclose(nrm[:, 2], 0.0)                                         # vertical walls
    assert np.isclose(0.5 * np.linalg.norm(nrm, axis=1).sum(), (16 + 8) * 1.0)  # perimeter x height
    cen = tri.mean(axis=1)
    # outer wall normals point away from the square's centre, the hole's walls towards it (away from the solid)
    out = np.einsum("ij,ij->i", nrm[:8, :2], cen[:8, :2] - 2.0)
    inn = np.einsum("ij,ij->i", nrm[8:, :2], cen[8:, :2] - 2.0)
    assert (out > 0).all() and (inn < 0).all()
    V, F = svsdf_amd.outline_extrude([outer, hole], z0=-0.5, z1=0.5, caps=True)
    assert V.shape == (16, 3) and len(F) == 16 + 2 * 8                         # 8 + 2 bridge vertices - 2 cap triangles, twice
    _closed_surface_checks(V, F, 16.0 - 4.0, 1.0)


def test_caps_on_curved_loops_holes_and_reference_outlines(built):
    """Closed surface on loops with many collinear / nearly collinear vertices: a disc with two holes, two separate
    components, and the z = 0 outlines of the reference's meshes (star: concave, sdArc: 754 vertices)."""
    import numpy as np
    import svsdf_amd
    from svsdf_amd import workload
    th = np.linspace(0.0, 2.0 * np.pi, 181)[:-1]
    circ = lambda cx, cy, r, sgn: np.c_[cx + r * np.cos(sgn * th), cy + r * np.sin(sgn * th)]
    area = lambda lp: 0.5 * np.sum(lp[:, 0] * np.roll(lp[:, 1], -1) - np.roll(lp[:, 0], -1) * lp[:, 1])
    loops = [circ(0, 0, 3.0, +1), circ(-1.2, 0.3, 0.8, -1), circ(1.1, -0.4, 0.9, -1), circ(8.0, 1.0, 1.5, +1)]
    V, F = svsdf_amd.outline_extrude(loops, z0=0.0, z1=2.0)
    _closed_surface_checks(V, F, sum(area(lp) for lp in loops), 2.0)
    for name in ("star", "sdArc", "sdHorseshoe"):
        lp = workload.mesh_outline(name)
        if area(lp) < 0:
            lp = lp[::-1].copy()
        V, F = svsdf_amd.outline_extrude([lp], z0=-0.5, z1=0.5)
        _closed_surface_checks(V, F, area(lp), 1.0)
