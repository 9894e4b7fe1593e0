"""Query-point producer: the product's host code against the oracle on the
reference's own demo maps (src/plan_manager/pcds/map_*.pcd, committed as data in
tests/golden/reference_assets.json), and -- on the GPU -- the whole "plumbing" chain of BASELINE
config C1: real map -> voxel centres around the waypoints -> cost + gradient, against the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import orc

ASSETS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_assets.json")))


def _scenario(name, N=8):
    from svsdf_amd import workload
    sc = ASSETS["scenarios"][name]
    q = workload.waypoints(sc["start"][:2], sc["end"][:2], N)
    halfbd = np.full(3, sc["kernel_size"] * sc["occupancy_resolution"] / 3.0)  # plan_manager.cpp:57-59,165
    return sc, q, halfbd


@pytest.mark.parametrize("name", ["star", "sdHorseshoe", "sdHeart"])
def test_producer_matches_oracle_on_reference_maps(built, name):
    import svsdf_amd
    cloud = np.array(ASSETS["maps"][name], dtype=np.float32)
    sc, q, halfbd = _scenario(name)
    m = svsdf_amd.OccupancyMap(cloud, sc["occupancy_resolution"], 1)
    pts = m.gather(q, halfbd)
    opts, dims = orc.map_points(cloud, q, halfbd, sc["occupancy_resolution"], 1)
    info = m.info()
    assert info["dims"] == dims
    assert len(pts) == len(opts) > 20
    np.testing.assert_array_equal(pts, opts)
    # every produced point is a voxel centre holding at least one cloud point
    res = sc["occupancy_resolution"]
    cells = {tuple(np.floor((c - info["bmin"]) / res).astype(int)) for c in cloud.astype(np.float64)}
    for p in pts:
        cell = tuple(np.round((p - info["bmin"]) / res - 0.5).astype(int))
        assert cell in cells


def test_pcd_reader_roundtrip(built, tmp_path):
    import svsdf_amd
    cloud = np.array(ASSETS["maps"]["star"], dtype=np.float32)
    path = tmp_path / "map.pcd"
    with open(path, "w") as f:   # header layout of src/plan_manager/pcds/map_star.pcd:1-11
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
                f"COUNT 1 1 1\nWIDTH {len(cloud)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(cloud)}\nDATA ascii\n")
        for p in cloud:
            f.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
    back = svsdf_amd.OccupancyMap.read_pcd(str(path))
    np.testing.assert_array_equal(back, cloud)
    with pytest.raises(svsdf_amd.SvsdfError):
        svsdf_amd.OccupancyMap.read_pcd(str(tmp_path / "missing.pcd"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["star", "sdHorseshoe", "sdHeart"])
def test_c1_plumbing_real_map_to_gradient(built, name):
    """run_<shape>.launch plumbing without ROS: demo map -> query points -> callback (x, g)."""
    import svsdf_amd
    from svsdf_amd import workload
    cloud = np.array(ASSETS["maps"][name], dtype=np.float32)
    sc, q, halfbd = _scenario(name)
    pts = svsdf_amd.OccupancyMap(cloud, sc["occupancy_resolution"], 1).gather(q, halfbd)
    hs, ts = workload.states(sc["start"][:2], sc["end"][:2])
    T = np.full(len(q) + 1, sc["inittime"])
    x = workload.x_from(q, T, svsdf_amd.backward_T)
    kw = dict(safety_hor=sc["safety_hor"], weight_p=sc["weight_p"], rho=sc["rho"], poly_params=sc["poly_params"],
              head_state=hs, tail_state=ts)
    ctx = svsdf_amd.SvsdfContext(shape=svsdf_amd.shape_id_from_inputdata(sc["inputdata"]), device=0, **kw)
    ctx.set_points(pts)
    f, g = ctx.lmbm_evaluate(x)
    o = orc.Oracle(name, **kw)
    fo, go, _ = o.cost_function(pts, x, nthreads=os.cpu_count() or 1)
    assert abs(f - fo) <= 1e-7 * abs(fo)
    assert np.linalg.norm(g - go) <= 1e-5 * np.linalg.norm(go)
