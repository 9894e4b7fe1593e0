"""The integer quotients k_solve's shared ladder takes through the hardware reciprocal (descend_from_seed, svsdf_kernels.hpp):
lanes per open ladder  wd = floor(64 / n)  as  int(64.5f * rcp(n)),  and a lane's ladder  a = floor(lane / wd)  as
int((lane + 0.5f) * rcp(wd)).  v_rcp_f32 is accurate to 1 ulp; the formulas must hold for any reciprocal within a few ulp."""
import numpy as np


def _rcp_variants(v):
    r = np.float32(1.0) / np.float32(v)
    out = [r]
    lo = hi = r
    for _ in range(3):
        lo = np.nextafter(lo, np.float32(0.0), dtype=np.float32)
        hi = np.nextafter(hi, np.float32(2.0), dtype=np.float32)
        out += [lo, hi]
    return out


def test_lanes_per_open_ladder():
    for n in range(1, 65):
        for r in _rcp_variants(n):
            assert int(np.float32(64.5) * r) == 64 // n, (n, r)


def test_ladder_of_a_lane():
    for wd in range(1, 65):
        for r in _rcp_variants(wd):
            for lane in range(64):
                assert int((np.float32(lane) + np.float32(0.5)) * r) == lane // wd, (wd, lane, r)
