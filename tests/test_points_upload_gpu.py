"""svsdf_set_points / svsdf_set_points_device plan the cloud on the device (bounding box -> Morton keys -> radix sort
-> stripe gather).  The order must be exactly the host planner's (svsdf_shard_plan: the plan every rank and every
device of a multi-device context agrees on), whatever the cloud looks like."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _clouds():
    rng = np.random.default_rng(11)
    a = np.zeros((5000, 3)); a[:, :2] = rng.uniform(-30, 80, (5000, 2)); a[:, 2] = rng.uniform(-1, 1, 5000)
    b = np.repeat(a[:40], 50, axis=0)                                     # heavy duplicates: ties broken by input index
    c = np.zeros((3000, 3)); c[:, 0] = np.linspace(0, 10, 3000)          # collinear (degenerate y extent)
    d = np.zeros((1, 3)); d[0, :2] = (3.0, -4.0)                         # single point
    e = np.tile(np.array([[1.5, 2.5, 0.0]]), (257, 1))                   # all points identical (extent 0)
    f = np.zeros((70001, 3)); f[:, :2] = rng.normal(0, 1e-9, (70001, 2)) + 5.0   # sub-quantum extent
    return dict(random=a, duplicates=b, collinear=c, single=d, identical=e, tiny=f)


@pytest.mark.parametrize("name", ["random", "duplicates", "collinear", "single", "identical", "tiny"])
def test_device_plan_equals_host_plan(built, name):
    import svsdf_amd
    pts = _clouds()[name]
    for rank, ws, flags in ((0, 1, 0), (1, 3, 0), (6, 8, 0), (0, 1, svsdf_amd.FLAG_KEEP_INPUT_ORDER),
                            (2, 5, svsdf_amd.FLAG_KEEP_INPUT_ORDER)):
        c = svsdf_amd.SvsdfContext(shape="star", device=0, rank=rank, world_size=ws, flags=flags)
        c.set_points(pts)
        want = svsdf_amd.shard_plan(pts, rank, ws, flags)
        np.testing.assert_array_equal(c.shard_indices(), want)
        assert c.num_points() == len(want)
        c.close()


def test_set_points_device_equals_host_upload(built):
    import torch
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C2", P=30000, minco=svsdf_amd.minco_coeffs)
    kw = dict(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
              head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    a = svsdf_amd.SvsdfContext(**kw)
    a.set_points(w["points"])
    b = svsdf_amd.SvsdfContext(**kw)
    t = torch.from_numpy(np.ascontiguousarray(w["points"])).to("cuda:0")
    torch.cuda.synchronize()
    b.set_points_device(t.data_ptr(), len(w["points"]))
    np.testing.assert_array_equal(a.shard_indices(), b.shard_indices())
    ra, rb = a.eval_penalty(w["coeffs"], w["T"]), b.eval_penalty(w["coeffs"], w["T"])
    assert ra[0] == rb[0]
    np.testing.assert_array_equal(ra[2], rb[2])
    # a multi-device context fed from device memory
    g = svsdf_amd.SvsdfContext(devices=[0, 0], **{k: v for k, v in kw.items() if k != "device"})
    g.set_points_device(t.data_ptr(), len(w["points"]))
    rg = g.eval_penalty(w["coeffs"], w["T"])
    assert abs(rg[0] - ra[0]) <= 1e-12 * abs(ra[0])
    assert g.stats()["setup_ms"] > 0


def test_nonfinite_points_are_rejected(built):
    import svsdf_amd
    pts = np.zeros((1000, 3)); pts[:, :2] = np.random.default_rng(0).uniform(0, 10, (1000, 2))
    for bad in (np.nan, np.inf, -np.inf):
        q = pts.copy(); q[517, 1] = bad
        c = svsdf_amd.SvsdfContext(shape="star", device=0)
        c.set_points(pts)
        with pytest.raises(svsdf_amd.SvsdfError, match="non-finite"):
            c.set_points(q)
        assert c.num_points() == 0            # a failed upload leaves no (stale) cloud behind
        c.set_points(pts)                     # still usable
        assert c.num_points() == 1000
        c.close()
