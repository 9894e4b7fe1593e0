"""CPU study (round 4): chunk evaluations of k_round's seed scan per GSIP sample -- list order (today) vs best-first order
(ascending lower bound, stop at the first bound above the running minimum).  Pose table, chunk circles and the candidate
list are rebuilt in numpy like k_prep / round_point build them; the shape values come from the oracle's shape SDF.
usage: python tools/experiments/scan_order.py [config] [n_points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np
import svsdf_amd
from svsdf_amd import workload
from oracle import orc

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 300
RB = {"sdHorseshoe": float(np.hypot(1.7, 1.55)), "star": 2.8, "sdHeart": 4.0 * (np.sqrt(0.25 ** 2 + 0.75 ** 2) + np.sqrt(2) / 4)}
w = workload.make(cfg, P=20000, minco=svsdf_amd.minco_coeffs)
o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], head_state=w["head_state"], tail_state=w["tail_state"])
o.set_traj(w["coeffs"], w["T"])
dur = o.duration()
tk = []
t = 0.0
while t <= dur:
    tk.append(t); t += 0.15
tk = np.array(tk)
K = len(tk)
pose = np.array([o.pos(t) for t in tk])            # x, y, yaw
cs, sn = np.cos(pose[:, 2]), np.sin(pose[:, 2])
nch = (K + 7) // 8
R = RB[w["shape"]] + 1e-6
ccx = np.zeros(nch); ccy = np.zeros(nch); crb = np.zeros(nch)
for c in range(nch):
    p = pose[8 * c:8 * c + 8, :2]
    cen = 0.5 * (p.min(0) + p.max(0))
    ccx[c], ccy[c] = cen
    crb[c] = np.linalg.norm(p - cen, axis=1).max() + R
sdf, ts, _ = o.query(w["points"], nthreads=os.cpu_count())
interior = w["points"][sdf < 0][:npts]
rng = np.random.default_rng(1)
print(f"{cfg}: K {K}, chunks {nch}, interior points used {len(interior)}")
for r in (10.0, 6.0, 3.0, 1.5, 0.7, 0.3):
    tot_list = tot_best = tot_cand = 0
    nq = 0
    for p in interior:
        th0 = rng.uniform(0, 2 * np.pi)
        ths = th0 + np.arange(21) * 0.3
        q = np.column_stack([p[0] + r * np.cos(ths), p[1] + r * np.sin(ths)])
        # candidate list of the round (round_point): chunks with |p - c| - r - rb <= U, U = min_c(|p - c| + rb) + r
        dpc = np.hypot(p[0] - ccx, p[1] - ccy)
        U = (dpc + crb).min() + r
        cand = np.nonzero(dpc - r - crb <= U)[0]
        tot_cand += len(cand)
        for qq in q:
            dx, dy = qq[0] - pose[:, 0], qq[1] - pose[:, 1]
            rel = np.column_stack([cs * dx + sn * dy, -sn * dx + cs * dy])
            val = o.shape_eval(rel)                                     # all table values of this query
            cmin = np.array([val[8 * c:8 * c + 8].min() for c in range(nch)])
            d = np.hypot(qq[0] - ccx[cand], qq[1] - ccy[cand])
            lb = d - crb[cand]
            # today: nearest centre first, then list order with the running minimum
            first = cand[np.argmin(d)]
            best = cmin[first]; n1 = 1
            for c, l in zip(cand, lb):
                if c == first: continue
                if l <= best:
                    n1 += 1; best = min(best, cmin[c])
            # best-first: ascending lower bound, stop when the bound exceeds the running minimum
            order = np.argsort(lb)
            best2 = np.inf; n2 = 0
            for k in order:
                if lb[k] > best2: break
                n2 += 1; best2 = min(best2, cmin[cand[k]])
            assert abs(best - best2) < 1e-12
            tot_list += n1; tot_best += n2; nq += 1
    print(f"  r = {r:5.1f}: candidates per point {tot_cand / len(interior):5.1f}   chunk evaluations per sample: list order {tot_list / nq:5.2f}   best-first {tot_best / nq:5.2f}")

# per-pose bound |q - x_k| - R against the running minimum: poses that would have to be evaluated if the scan pruned
# single poses (packed 8 at a time) instead of chunks of 8
print("per-pose pruning (nearest chunk first, then the poses of the candidate chunks whose own bound reaches the minimum so far):")
for r in (10.0, 3.0, 0.7):
    tot = nq = tot_final = 0
    for p in interior[:80]:
        th0 = rng.uniform(0, 2 * np.pi)
        ths = th0 + np.arange(21) * 0.3
        q = np.column_stack([p[0] + r * np.cos(ths), p[1] + r * np.sin(ths)])
        dpc = np.hypot(p[0] - ccx, p[1] - ccy)
        U = (dpc + crb).min() + r
        cand = np.nonzero(dpc - r - crb <= U)[0]
        for qq in q:
            dx, dy = qq[0] - pose[:, 0], qq[1] - pose[:, 1]
            rel = np.column_stack([cs * dx + sn * dy, -sn * dx + cs * dy])
            val = o.shape_eval(rel)
            d = np.hypot(qq[0] - ccx[cand], qq[1] - ccy[cand])
            first = cand[np.argmin(d)]
            best0 = val[8 * first:8 * first + 8].min()
            ks = np.concatenate([np.arange(8 * c, min(8 * c + 8, K)) for c in cand if c != first])
            lbk = np.hypot(dx[ks], dy[ks]) - R
            surv = ks[lbk <= best0]
            tot += 8 + len(surv)
            fin = min(best0, val[surv].min()) if len(surv) else best0
            tot_final += 8 + int((lbk <= fin).sum())
            nq += 1
    print(f"  r = {r:5.1f}: table evaluations per sample: {tot / nq:5.1f} (bound after the first chunk), {tot_final / nq:5.1f} (bound = final minimum)")
