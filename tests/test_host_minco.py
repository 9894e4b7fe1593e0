"""Host logic of the product library (MINCO S3NU forward, tau/T maps) against the oracle."""
import numpy as np

from oracle import orc


def _case(N, seed):
    rng = np.random.default_rng(seed)
    hs = np.zeros((3, 3)); ts = np.zeros((3, 3))
    hs[:, 0] = rng.uniform(-5, 5, 3); hs[:, 1] = rng.uniform(-1, 1, 3); hs[:, 2] = rng.uniform(-1, 1, 3)
    ts[:, 0] = rng.uniform(20, 30, 3); ts[:, 1] = rng.uniform(-1, 1, 3)
    q = np.cumsum(rng.uniform(0.5, 2.0, (N - 1, 3)), axis=0)
    T = rng.uniform(0.6, 3.5, N)
    return hs, ts, q, T


def test_minco_coeffs_match_oracle(built):
    import svsdf_amd
    for N in (1, 2, 3, 8, 16, 32):
        hs, ts, q, T = _case(N, N)
        a = svsdf_amd.minco_coeffs(hs, ts, q, T)
        b = orc.minco_coeffs(hs, ts, q, T)
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


def test_minco_interpolates_and_is_continuous(built):
    import svsdf_amd
    N = 8
    hs, ts, q, T = _case(N, 3)
    c = svsdf_amd.minco_coeffs(hs, ts, q, T).reshape(N, 6, 3)
    for i in range(N - 1):
        s = T[i] ** np.arange(6)
        np.testing.assert_allclose(s @ c[i], q[i], atol=1e-9)
        np.testing.assert_allclose(c[i + 1][0], q[i], atol=1e-9)
        # velocity .. snap continuity (d = 1..4)
        for d in range(1, 5):
            k = np.arange(d, 6)
            f = np.array([np.prod(np.arange(kk - d + 1, kk + 1)) for kk in k], dtype=float)
            end = (f * T[i] ** (k - d)) @ c[i][d:]
            start = f[0] * c[i + 1][d]
            np.testing.assert_allclose(end, start, atol=1e-7)
    np.testing.assert_allclose(c[0][0], hs[:, 0]); np.testing.assert_allclose(c[0][1], hs[:, 1])
    np.testing.assert_allclose(2 * c[0][2], hs[:, 2])


def test_tau_maps(built):
    import svsdf_amd
    tau = np.linspace(-4, 4, 33)
    np.testing.assert_array_equal(svsdf_amd.forward_T(tau), orc.forward_T(tau))
    T = svsdf_amd.forward_T(tau)
    np.testing.assert_array_equal(svsdf_amd.backward_T(T), orc.backward_T(T))
