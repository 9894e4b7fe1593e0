"""CPU-only study: site executions of k_solve's descent under different lane-sharing schemes, replayed from the oracle's
trace of every gradientDescent call (passes and accepted ladder index per pass) -- which restructuring is worth building?

usage: python tools/experiments/wave_model.py [config] [points]
Schemes (a wave holds Q = 64 / G queries and runs them to completion, like k_solve):
  fixed    G lanes per query for its own ladder, the wave runs the slowest query's steps (rounds 1-2)
  shared   the 64 lanes dealt out evenly to the ladders still open in the pass (round 3, SVSDF_ELASTIC)
  pool     no pass barrier: every step serves whatever each query needs next (its derivative tasks or the next
           candidates of its ladder) from one pool of 64 lanes -- the "request pool" form
  bound    total lane-evaluations / 64: what perfect packing with refill from the queue could reach
Counts are evaluation-site executions per solve (scan layers 2-4 + derivative + ladder); compare `fixed` / `shared` with
the in-kernel counters (profiles/r03_site_stats_*): C3 measured 7.26 (fixed, G = 4), 4.97 (shared, 4), 4.40 (shared, 2)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd"), os.path.join(ROOT, "oracle")]
import orc  # noqa: E402
from svsdf_amd import workload  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
w = workload.make(cfg, P=P, minco=orc.minco_coeffs)
o = orc.Oracle(w["shape"], poly_params=w["poly_params"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
               polygon=w["polygon"])
o.set_traj(w["coeffs"], w["T"])
cap = 40 * P
buf = np.zeros((cap, 32), dtype=np.uint8)
L = o.L
L.orc_set_gd_trace.argtypes = [C.c_void_p, C.c_size_t]
L.orc_gd_trace_count.restype = C.c_size_t
L.orc_set_gd_trace(buf.ctypes.data_as(C.c_void_p), cap)
# Morton-ish order (the product sorts its cloud): sort by a coarse cell, then evaluate single-threaded so that the trace
# order is the evaluation order (a point's main solve, then its GSIP samples)
pts = w["points"]
key = (np.floor(pts[:, 0] / 0.5).astype(np.int64) << 20) + np.floor(pts[:, 1] / 0.5).astype(np.int64)
o.penalty(pts[np.argsort(key, kind="stable")], nthreads=1)
n = min(int(L.orc_gd_trace_count()), cap)
L.orc_set_gd_trace(None, 0)
rec = buf[:n]
print(f"{cfg}: {P} points, {n} descents, passes/descent {rec[:, 0].mean():.2f}, max {rec[:, 0].max()}")


def ladders(r):
    """per pass: accepted index (1..29) or 30 = none accepted (29 candidates tried in vain)"""
    return [int(r[k]) if r[k] else 30 for k in range(1, int(r[0]) + 1)]


def fixed(batch, G):
    ex = 3 * -(-21 // G)
    seqs = [ladders(r) for r in batch]
    for p in range(max(len(s) for s in seqs)):
        run = [s[p] for s in seqs if len(s) > p]
        ex += -(-(3 if p == 0 else 2) // G)
        ex += max(-(-min(j, 29) // G) for j in run)
    return ex


def shared(batch, G):
    ex = 3 * -(-21 // G)
    seqs = [ladders(r) for r in batch]
    for p in range(max(len(s) for s in seqs)):
        run = [s[p] for s in seqs if len(s) > p]
        ex += -(-(3 if p == 0 else 2) // G)
        j0, open_ = 1, list(run)
        while open_:
            wd = min(32, 64 // len(open_))
            ex += 1
            open_ = [j for j in open_ if not (j0 <= j < j0 + wd) and j0 + wd <= 29]
            j0 += wd
    return ex


def pool(batch, G):
    ex = 3 * -(-21 // G)
    # state per query: (pass index, phase 0 = derivative / 1 = ladder, next candidate)
    st = [[0, 0, 1, ladders(r)] for r in batch]
    while st:
        ex += 1
        der = [q for q in st if q[1] == 0]
        lad = [q for q in st if q[1] == 1]
        lanes = 64 - sum(3 if q[0] == 0 else 2 for q in der)
        wd = max(1, min(32, lanes // len(lad))) if lad else 0
        for q in lad:
            j = q[3][q[0]]
            if q[2] <= j < q[2] + wd:
                q[0] += 1; q[1] = 0; q[2] = 1
            elif q[2] + wd > 29:
                q[0] = len(q[3])          # failed ladder: the descent stops
            else:
                q[2] += wd
        for q in der:
            q[1] = 1
        st = [q for q in st if q[0] < len(q[3])]
    return ex


def bound(batch):
    ev = 0
    for r in batch:
        ev += 63
        for p, j in enumerate(ladders(r)):
            ev += (3 if p == 0 else 2) + min(j, 29)
    return ev / 64.0


for G in (4, 2, 1):
    Q = 64 // G
    nb = n // Q
    tot = {"fixed": 0, "shared": 0, "pool": 0, "bound": 0.0}
    for b in range(nb):
        batch = rec[b * Q:(b + 1) * Q]
        tot["fixed"] += fixed(batch, G)
        tot["shared"] += shared(batch, G)
        tot["pool"] += pool(batch, G)
        tot["bound"] += bound(batch)
    per = {k: v / (nb * Q) for k, v in tot.items()}
    print(f"  G = {G}: site executions per solve  " + "  ".join(f"{k} {v:.2f}" for k, v in per.items()))
