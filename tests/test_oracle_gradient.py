"""Finite-difference check of the oracle's assembled gradient (the chain BEO:797-863 -> MINCO adjoint
MNC:584-654 -> tau/xi maps BEO:268-314) on query points whose SVSDF is smooth in the trajectory
parameters (exterior and active: 0 < sdf < safety_hor, away from the smoothed-L1 knees)."""
import os

import numpy as np

from oracle import orc

NT = min(os.cpu_count() or 1, 16)


def _case():
    start, end = (4.3987178802490234, 4.7499313354492188), (20.23274040222168, 64.403488159179688)
    N = 8
    u = (np.arange(N - 1) + 1.0) / N
    q = np.zeros((N - 1, 3))
    q[:, 0] = start[0] * (1 - u) + end[0] * u + 2.0 * np.sin(np.pi * u) * np.sin(2 * np.pi * 1.5 * u)
    q[:, 1] = start[1] * (1 - u) + end[1] * u
    q[:, 2] = 0.6 * np.sin(2 * np.pi * u)
    hs, ts = np.zeros((3, 3)), np.zeros((3, 3))
    hs[:2, 0], ts[:2, 0] = start, end
    T = np.array([2.1, 2.6, 2.4, 2.9, 2.2, 2.5, 2.7, 2.3])
    return hs, ts, q, T


def test_cost_function_gradient_matches_finite_differences():
    hs, ts, q, T = _case()
    o = orc.Oracle("star", safety_hor=0.7, weight_p=60.0, rho=3.8, head_state=hs, tail_state=ts)
    coeffs = orc.minco_coeffs(hs, ts, q, T)
    o.set_traj(coeffs, T)
    rng = np.random.default_rng(7)
    cand = np.zeros((4000, 3))
    idx = rng.integers(0, len(q), 4000)
    cand[:, :2] = q[idx, :2] + rng.uniform(-5.0, 5.0, (4000, 2))
    sdf, tstar, _ = o.query(cand, nthreads=NT)
    ok = (sdf > 0.1) & (sdf < 0.55) & (tstar > 1.0) & (tstar < T.sum() - 1.0)
    pts = cand[ok][:60]
    assert len(pts) >= 30
    x = np.concatenate([orc.backward_T(T), q.ravel()])
    f0, g, c3 = o.cost_function(pts, x, nthreads=NT)
    assert c3[0] > 0 and abs(c3[2] - f0) < 1e-12 * abs(f0)
    comps = rng.choice(len(x), 12, replace=False)
    for i in comps:
        h = 1e-6
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        fp, _, _ = o.cost_function(pts, xp, nthreads=NT)
        fm, _, _ = o.cost_function(pts, xm, nthreads=NT)
        fd = (fp - fm) / (2 * h)
        assert abs(fd - g[i]) <= 2e-4 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_penalty_accumulates_and_prefix_rule():
    """gradT[j] receives gdT of every point whose t* lies in a LATER piece (BEO:859-862)."""
    hs, ts, q, T = _case()
    o = orc.Oracle("star", head_state=hs, tail_state=ts)
    o.set_traj(orc.minco_coeffs(hs, ts, q, T), T)
    # one active exterior point near piece 5
    p = np.array([[q[4, 0] + 2.2, q[4, 1] + 0.3, 0.0]])
    sdf, tstar, _ = o.query(p)
    c, gT, gC = o.penalty(p)
    piece = int(np.searchsorted(np.cumsum(T), tstar[0]))
    if c > 0:
        assert np.all(gT[piece:] == 0.0) and np.all(gT[:piece] == gT[0])
        rows = gC.reshape(len(T), 6, 3)
        assert np.all(rows[np.arange(len(T)) != piece] == 0.0)
    c2, gT2, gC2 = o.penalty(p, cost0=1.5, gradT0=np.ones(len(T)), gradC0=np.full((6 * len(T), 3), 2.0))
    assert abs(c2 - 1.5 - c) < 1e-12 and np.allclose(gT2 - 1.0, gT) and np.allclose(gC2 - 2.0, gC)
