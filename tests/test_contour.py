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
    return dict(nodes=nodes, dense=dense, open=open_chains, loops=loops, sx=sx, sy=sy, maxres=maxres)


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
    # same cells marched where the zero set is -> the very same vertices (crossing points depend on two node values only)
    assert sorted(band["loops"]) == sorted(dense["loops"])
    assert band["sx"] == pytest.approx(dense["sx"], abs=1e-9) and band["sy"] == pytest.approx(dense["sy"], abs=1e-9)


def test_outline_extrusion_is_a_wall_with_outward_normals(built):
    """svsdf_outline_extrude (host only): a counter-clockwise square with a clockwise square hole."""
    import numpy as np
    import svsdf_amd
    outer = np.array([[0, 0], [4, 0], [4, 4], [0, 4]], dtype=float)            # inside on the left
    hole = np.array([[1, 1], [1, 3], [3, 3], [3, 1]], dtype=float)             # clockwise
    V, F = svsdf_amd.outline_extrude([outer, hole], z0=-0.5, z1=0.5)
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
