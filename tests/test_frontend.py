"""SweptVolumeManager::checkSubSWCollision (sw_manager.hpp:1171-1211) and
BasicShape::initShape (Shape.hpp:386-430).

CPU: the C oracle against the golden vectors of the independent pure-Python restatement
(tests/golden/golden_frontend.json), analytic anchors, and the reference's own shape meshes.
GPU: the HIP kernels (through the C ABI) against the oracle -- boolean outputs, so the bar is exact equality."""
import json
import math
import os

import numpy as np
import pytest

from oracle import orc

HERE = os.path.dirname(__file__)
GOLD = json.load(open(os.path.join(HERE, "golden", "golden_frontend.json")))["cases"]
ASSETS = json.load(open(os.path.join(HERE, "golden", "reference_assets.json")))
ANALYTIC = [s for s in orc.SHAPES if s != "Polygon"]


def _oracle(case):
    return orc.Oracle(case["shape"], poly_params=case["poly_params"], polygon=case["polygon"])


# ---------------------------------------------------------------- CPU: oracle vs golden / anchors
@pytest.mark.parametrize("case", GOLD, ids=[c["shape"] for c in GOLD])
def test_oracle_collision_matches_python_restatement(case):
    o = _oracle(case)
    got = [o.check_sub_sw_collision(e["father"], e["child"], e["points"]) for e in case["edges"]]
    assert got == [e["free"] for e in case["edges"]]
    assert any(got) and not all(got)  # both outcomes are exercised


@pytest.mark.parametrize("case", [c for c in GOLD if "kernels" in c], ids=[c["shape"] for c in GOLD if "kernels" in c])
def test_oracle_shape_kernels_match_python_restatement(case):
    o = _oracle(case)
    m, b, yaws, n = o.shape_kernels(case["kernel_size"], case["kernel_count"], case["resolution"], case["safemargin"])
    assert n == case["loop_count"]
    np.testing.assert_array_equal(m, np.array(case["kernels"], dtype=bool))
    np.testing.assert_array_equal(yaws, np.array(case["yaws"]))
    # generateByteKernel (Shape.hpp:194-216): or_mask[b % 8] = 0x80 >> (b % 8)  == numpy's big-endian packbits
    np.testing.assert_array_equal(b, np.packbits(m, axis=2, bitorder="big"))


def test_interpolation_loop_has_50_steps():
    # `for (double kt = 0.0; kt <= 1.0; kt += 0.02)` (sw_manager.hpp:1189): 50 accumulated adds of 0.02 give
    # 1.0000000000000004 > 1, so the loop runs kt = 0 ... 0.98 and the child pose itself is never tested.
    kt, n, last = 0.0, 0, 0.0
    while kt <= 1.0:
        last = kt
        kt += 0.02
        n += 1
    assert n == 50 and abs(last - 0.98) < 1e-12
    # consequence, reproduced by the oracle: a point strictly inside the star at the child pose only is "free"
    o = orc.Oracle("star")
    assert orc.shape_sdf("star", 0.0, 2.79) < 0                                          # just below the upper tip
    assert o.check_sub_sw_collision([0, 0, 0], [10.0, 0, 0], [[10.0, 2.79]])             # inside at the child pose only
    assert not o.check_sub_sw_collision([0, 0, 0], [10.0, 0, 0], [[9.8, 2.79]])          # tip passes here at kt = 0.98


def test_collision_anchors_star():
    o = orc.Oracle("star")
    # pure translation along x: a point 0.05 above the upper tip (0, 2.8) is never touched, one on the axis is
    assert o.check_sub_sw_collision([0, 0, 0], [1, 0, 0], [[0.0, 2.85]])
    assert not o.check_sub_sw_collision([0, 0, 0], [1, 0, 0], [[0.5, 0.0]])
    # the swept tip passes x = 0.5 at kt = 0.5: (0.5, 2.79) is inside the tip there and nowhere else on the grid
    assert not o.check_sub_sw_collision([0, 0, 0], [1, 0, 0], [[0.5, 2.79]])
    # rotation in place by 36 deg sweeps the tip over the point at polar (2.7, 90 deg + 18 deg); the static star misses it
    ang = math.radians(108.0)
    p = [[2.7 * math.cos(ang), 2.7 * math.sin(ang)]]
    assert o.check_sub_sw_collision([0, 0, 0], [0, 0, 0], p)
    assert not o.check_sub_sw_collision([0, 0, 0], [0, 0, math.radians(36.0)], p)
    # no obstacle points -> free
    assert o.check_sub_sw_collision([0, 0, 0], [1, 1, 0.3], np.zeros((0, 2)))


@pytest.mark.parametrize("shape", ["star", "sdHorseshoe", "sdHeart", "sdCutDisk"])
def test_zero_yaw_kernel_covers_reference_mesh(shape):
    """Every vertex of the reference's own outline mesh (shapes/<name>.obj) lies on the zero level set, so the
    kernel cell nearest to it (resolution 0.25, margin = resolution/2 like Shape.hpp:399) must be occupied in
    the yaw = 0 kernel -- which is index kernel_count/2 (`zero_yaw_ind`, Shape.hpp:390)."""
    verts = np.array(ASSETS["shapes"][shape])[:, :2]
    res, ks, K = 0.25, 81, 18
    o = orc.Oracle(shape)
    m, _, yaws, n = o.shape_kernels(ks, K, res, res / 2)
    k0 = K // 2
    assert abs(yaws[k0]) < 1e-12
    side = (ks - 1) // 2
    # row vector times R_obj(0) = identity: cell (a, b) sits at (res*(a-side), res*(b-side)) in the shape frame
    a = np.rint(verts[:, 0] / res).astype(int) + side
    b = np.rint(verts[:, 1] / res).astype(int) + side
    ok = (a >= 0) & (a < ks) & (b >= 0) & (b < ks)
    assert ok.all()
    # the nearest cell centre is within res/sqrt(2) of a boundary vertex; the SDF is 1-Lipschitz, so allow one ring
    hit = np.zeros(len(verts), dtype=bool)
    for da in (-1, 0, 1):
        for db in (-1, 0, 1):
            aa, bb = np.clip(a + da, 0, ks - 1), np.clip(b + db, 0, ks - 1)
            hit |= m[k0, aa, bb]
    assert hit.mean() > 0.99


def test_kernel_yaw_symmetry_star():
    # the star is mirror-symmetric in x: kernel(-yaw) is kernel(+yaw) mirrored in a (x index)
    o = orc.Oracle("star")
    m, _, yaws, _ = o.shape_kernels(17, 18, 1.0, 0.5)
    for k in range(1, 9):
        assert abs(yaws[9 + k] + yaws[9 - k]) < 1e-12
        np.testing.assert_array_equal(m[9 + k], m[9 - k][::-1, :])


# ---------------------------------------------------------------- GPU: HIP vs oracle
def _random_edges(rng, n_edges, max_pts, spread):
    fs = np.column_stack([rng.uniform(5, 25, n_edges), rng.uniform(5, 25, n_edges), rng.uniform(-math.pi, math.pi, n_edges)])
    cs = fs + np.column_stack([rng.integers(-1, 2, n_edges), rng.integers(-1, 2, n_edges), rng.uniform(-0.7, 0.7, n_edges)])
    pts = []
    for e in range(n_edges):
        n = int(rng.integers(0, max_pts + 1))
        pts.append(fs[e, :2] + rng.uniform(-spread, spread, (n, 2)))
    return fs, cs, pts


@pytest.mark.gpu
@pytest.mark.parametrize("shape", orc.SHAPES)
def test_hip_collision_matches_oracle_all_shapes(built, shape):
    import svsdf_amd
    rng = np.random.default_rng(orc.SHAPE_ID[shape] + 77)
    from svsdf_amd import workload
    kw = dict(polygon=workload.star_outline()) if shape == "Polygon" else {}
    ctx = svsdf_amd.SvsdfContext(shape=shape, device=0, **kw)
    o = orc.Oracle(shape, **kw)
    fs, cs, pts = _random_edges(rng, 300, 12, 7.0)
    got = ctx.check_sub_sw_collision(fs, cs, pts)
    want = np.array([o.check_sub_sw_collision(fs[e], cs[e], pts[e]) for e in range(len(fs))])
    np.testing.assert_array_equal(got, want)
    assert want.any() and not want.all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD, ids=[c["shape"] for c in GOLD])
def test_hip_frontend_matches_golden(built, case):
    import svsdf_amd
    ctx = svsdf_amd.SvsdfContext(shape=case["shape"], poly_params=case["poly_params"], polygon=case["polygon"], device=0)
    got = ctx.check_sub_sw_collision([e["father"] for e in case["edges"]], [e["child"] for e in case["edges"]],
                                     [e["points"] for e in case["edges"]])
    assert list(got) == [e["free"] for e in case["edges"]]
    if "kernels" in case:
        m, b, yaws, n = ctx.shape_kernels(case["kernel_size"], case["kernel_count"], case["resolution"], case["safemargin"])
        assert n == case["loop_count"]
        np.testing.assert_array_equal(m, np.array(case["kernels"], dtype=bool))
        np.testing.assert_array_equal(b, np.packbits(m, axis=2, bitorder="big"))
        np.testing.assert_array_equal(yaws, np.array(case["yaws"]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ANALYTIC)
def test_hip_shape_kernels_match_oracle_all_shapes(built, shape):
    import svsdf_amd
    ctx = svsdf_amd.SvsdfContext(shape=shape, poly_params=(0.3, -0.2, 25.0), device=0)
    o = orc.Oracle(shape, poly_params=(0.3, -0.2, 25.0))
    for ks, K, res, margin in ((17, 18, 1.0, 0.5), (21, 36, 0.7, 0.35)):
        m, b, yaws, n = ctx.shape_kernels(ks, K, res, margin)
        mo, bo, yo, no = o.shape_kernels(ks, K, res, margin)
        assert n == no
        np.testing.assert_array_equal(yaws, yo)
        # a cell may differ only if its SDF sits within rounding of the margin (device vs host sin/cos): none expected
        np.testing.assert_array_equal(m, mo)
        np.testing.assert_array_equal(b, bo)


@pytest.mark.gpu
def test_hip_collision_edge_cases(built):
    import svsdf_amd
    ctx = svsdf_amd.SvsdfContext(shape="star", device=0)
    # no edges, edges without points, a large ragged batch crossing the 64-point block chunks
    assert len(ctx.check_sub_sw_collision(np.zeros((0, 3)), np.zeros((0, 3)), [])) == 0
    got = ctx.check_sub_sw_collision([[0, 0, 0], [0, 0, 0]], [[1, 0, 0], [1, 1, 0.2]], [np.zeros((0, 2)), np.zeros((0, 2))])
    assert got.tolist() == [True, True]
    rng = np.random.default_rng(5)
    o = orc.Oracle("star")
    fs = np.array([[10.0, 10.0, 0.3]] * 4)
    cs = np.array([[11.0, 10.0, 0.5], [10.0, 11.0, -0.2], [11.0, 11.0, 0.3], [9.0, 9.0, 1.0]])
    far = fs[0, :2] + np.array([30.0, 0.0]) + rng.uniform(-1, 1, (1000, 2))       # all far away: free
    one = np.vstack([far[:777], [[10.4, 10.1]], far[777:]])                           # one colliding point in chunk 12
    ring = 10.0 + 3.5 * np.column_stack([np.cos(np.linspace(0, 6.28, 257)), np.sin(np.linspace(0, 6.28, 257))])
    pts = [far, one, ring, far[:1]]
    got = ctx.check_sub_sw_collision(fs, cs, pts)
    want = [o.check_sub_sw_collision(fs[e], cs[e], pts[e]) for e in range(4)]
    assert got.tolist() == want
    assert want[0] and not want[1]
    with pytest.raises(Exception):
        svsdf_amd.SvsdfContext(shape="Polygon", device=0).shape_kernels(17, 18, 1.0, 0.5)


@pytest.mark.gpu
def test_hip_astar_expansions_on_reference_map(built):
    """The call pattern of AstarGetSucc (front_end_Astar.hpp:192-241) on the reference's demo map: for a row of
    cells, the 9 neighbour edges with the obstacle cells inside the (kernel_size/2+1) box around each child."""
    import svsdf_amd
    cloud = np.array(ASSETS["maps"]["star"], dtype=np.float32)
    obst = np.unique(np.floor(cloud[:, :2]).astype(int), axis=0) + 0.5    # occupied cell centres at resolution 1
    ctx = svsdf_amd.SvsdfContext(shape="star", device=0)
    o = orc.Oracle("star")
    half = 17 // 2 + 1
    fs, cs, pts = [], [], []
    for cx in range(3, 28, 4):
        for cy in range(5, 70, 6):
            for i in (-1, 0, 1):
                for j in (-1, 0, 1):
                    child = np.array([cx + i + 0.5, cy + j + 0.5])
                    sel = obst[(np.abs(obst[:, 0] - child[0]) <= half) & (np.abs(obst[:, 1] - child[1]) <= half)]
                    fs.append([cx + 0.5, cy + 0.5, 0.2 * i])
                    cs.append([child[0], child[1], 0.2 * i + 0.35 * j])
                    pts.append(sel)
    got = ctx.check_sub_sw_collision(np.array(fs), np.array(cs), pts)
    want = np.array([o.check_sub_sw_collision(fs[e], cs[e], pts[e]) for e in range(len(fs))])
    np.testing.assert_array_equal(got, want)
    assert want.any() and not want.all()
