"""The unit of work of the whole path -- SDF at a time stamp (getSDFAtTimeStamp<false>, sw_manager.hpp:741-750: trajectory
pose, posEva2Rel, the shape's own offset / rotation, the shape SDF) -- evaluated on the device through the code the solve
kernels inline (svsdf_debug_sdf_at) against the oracle in device-arithmetic mode: bit for bit, every shape, random shape
offsets and rotations, generic piece durations (the reference's chain of subtractions) and dyadic ones.

Round 4: the device-arithmetic fuzz found sdRoundedCross at a shape rotation of -72.42 degrees one ulp off -- the oracle's
cos(yaw) / sin(yaw) pair is merged into one glibc sincos() by gcc -O3 (as in the reference's own build, Shape.hpp:289-292),
the product's host code called cos() and sin(); both call sincos() now.  This test pins that rotation."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
SHAPES = list(orc.SHAPES)


@pytest.mark.parametrize("shape", SHAPES)
def test_sdf_at_time_is_bit_identical(built, shape):
    import svsdf_amd
    from svsdf_amd import workload
    rng = np.random.default_rng(1000 + SHAPES.index(shape))
    poly = workload.star_outline() if shape == "Polygon" else None
    for trial, pp in enumerate([(-0.2825779765147789, 0.7125194778521649, -72.4229346429489), (0.0, 0.0, 0.0),
                                tuple(rng.uniform(-1, 1, 2)) + (float(rng.uniform(-180, 180)),),
                                tuple(rng.uniform(-1, 1, 2)) + (float(rng.uniform(-180, 180)),)]):
        if shape == "Polygon":
            pp = (0.0, 0.0, 0.0)          # Polygon::getonlySDF applies no offset (SHP:1448-1476)
        N = 5
        T = rng.uniform(0.4, 3.0, N) if trial % 2 == 0 else np.full(N, 2.5)     # generic (chain) / dyadic (cumulative)
        hs = np.zeros((3, 3)); ts = np.zeros((3, 3))
        hs[:2, 0] = rng.uniform(0, 10, 2); ts[:2, 0] = hs[:2, 0] + rng.uniform(-10, 10, 2)
        hs[2, 0] = rng.uniform(-3, 3); ts[2, 0] = rng.uniform(-3, 3)
        q = np.column_stack([np.linspace(hs[0, 0], ts[0, 0], N + 1)[1:-1] + rng.uniform(-2, 2, N - 1),
                             np.linspace(hs[1, 0], ts[1, 0], N + 1)[1:-1] + rng.uniform(-2, 2, N - 1),
                             rng.uniform(-2.5, 2.5, N - 1)])
        coeffs = svsdf_amd.minco_coeffs(hs, ts, q, T)
        kw = dict(poly_params=pp, polygon=poly, head_state=hs, tail_state=ts)
        ctx = svsdf_amd.SvsdfContext(shape=shape, device=0, **kw)
        o = orc.Oracle(shape, **kw)
        o.set_traj(coeffs, T)
        o.set_modes(1, 0)        # device-library trig, the reference's piece location
        n = 3000
        tt = rng.uniform(0.0, T.sum(), n)
        tt[:N + 1] = np.concatenate([[0.0], np.cumsum(T)])[:N + 1]       # piece boundaries (sequential sums)
        pos = np.array([o.pos(t)[:2] for t in tt])
        xy = pos + rng.normal(0, 3.0, (n, 2))
        xy[: n // 10] = pos[: n // 10]                                    # on the path: deep interior
        d = ctx.debug_sdf_at(coeffs, T, xy, tt)
        ref = np.array([o.sdf_at_time(x, y, t) for (x, y), t in zip(xy, tt)])
        assert d[0, 7] == (1.0 if trial % 2 == 0 else 0.0)                # the chain runs for generic durations only
        assert np.array_equal(d[:, 1], np.array([o.pos(t)[0] for t in tt])), (shape, pp, "pose x")
        bad = np.nonzero(d[:, 0] != ref)[0]
        assert len(bad) == 0, (shape, pp, len(bad), xy[bad[:3]], tt[bad[:3]], d[bad[:3], 0], ref[bad[:3]])
        ctx.close()
