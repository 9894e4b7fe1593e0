"""Why the gradient gate (1e-5) cannot hold on plateau trajectories: the oracle against ITSELF with another libm.

Found by differential fuzzing (profiles/r02_v4_fuzz_150.txt, seed 4242 case 93): the robot turns on the spot (start ==
end, one piece), the sector shape's SDF is constant in t over whole intervals, and getSDFofSweptVolume's strict-< argmin
(SWM:549-576, 1249-1325) is decided by the last bit of sin / cos.  The oracle evaluated with glibc's trig and with the
device library's trig then returns the same cost and sdf values but t* up to 0.9 s apart and coefficient gradients 6 %
apart -- exactly the deviation the HIP path shows against the glibc oracle, while it is bit-identical to the
device-trig oracle.  This test pins that property of the reference algorithm (CPU only)."""
import numpy as np

from oracle import orc


def test_rotation_on_the_spot_has_no_stable_argmin():
    rng = np.random.default_rng(7)
    start = np.array([11.0, 7.0])
    hs = np.zeros((3, 3)); ts = np.zeros((3, 3))
    hs[:2, 0] = start; ts[:2, 0] = start
    hs[2, 0] = -1.3; ts[2, 0] = 2.1                       # yaw only: rotation on the spot
    T = np.array([0.983])
    coeffs = orc.minco_coeffs(hs, ts, np.zeros((0, 3)), T)
    pts = np.zeros((300, 3))
    pts[:, :2] = start + rng.normal(0, 2.5, (300, 2))
    kw = dict(safety_hor=0.457, weight_p=60.0, rho=3.8, head_state=hs, tail_state=ts)
    a = orc.Oracle("sdPie", **kw); a.set_traj(coeffs, T)
    b = orc.Oracle("sdPie", **kw); b.set_traj(coeffs, T); b.set_modes(1, 0)   # device-library sin / cos / atan2
    ca, _, gCa, sdfa, tsa, _ = a.penalty(pts, nthreads=2, sum_mode=1, per_point=True)
    cb, _, gCb, sdfb, tsb, _ = b.penalty(pts, nthreads=2, sum_mode=1, per_point=True)
    assert abs(ca - cb) <= 1e-9 * abs(ca)                  # the cost is well defined ...
    np.testing.assert_allclose(sdfa, sdfb, rtol=0, atol=1e-9)
    moved = np.abs(tsa - tsb) > 1e-3                       # ... the argmin time is not
    assert moved.any()
    np.testing.assert_allclose(sdfa[moved], sdfb[moved], rtol=0, atol=1e-9)   # plateau: same value at both times
