"""CPU study (round 6): chunk STEPS of k_round's seed scan per GSIP sample -- today's circle test against an anchored
Lipschitz bound.  One step = what an 8-lane group does in one go: the 8 poses of one chunk, or ONE pose (the anchor) of
8 different chunks.

Anchored bound: every shape SDF is 1-Lipschitz, so for the poses k of chunk c and its anchor pose a (the chunk's middle)
  sdf_k(q) >= sdf_a(q) - (max_k |x_k - x_a| + max_k 2 |sin((yaw_k - yaw_a) / 2)| * |q - x_a|).
usage: python tools/experiments/anchor_chunk_bound.py [config] [n_points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np
from svsdf_amd import workload
from oracle import orc

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 200
RB = {"sdHorseshoe": float(np.hypot(1.7, 1.55)), "star": 2.8, "sdHeart": 4.0 * (np.sqrt(0.25 ** 2 + 0.75 ** 2) + np.sqrt(2) / 4)}
w = workload.make(cfg, P=20000, minco=orc.minco_coeffs)
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
CH = int(os.environ.get("CHUNK", "8"))
nch = (K + CH - 1) // CH
R = RB[w["shape"]] + 1e-6
ccx = np.zeros(nch); ccy = np.zeros(nch); crb = np.zeros(nch)
ka = np.zeros(nch, dtype=int); adx = np.zeros(nch); asm = np.zeros(nch)
for c in range(nch):
    k0, k1 = CH * c, min(CH * c + CH, K)
    p = pose[k0:k1, :2]
    cen = 0.5 * (p.min(0) + p.max(0))
    ccx[c], ccy[c] = cen
    crb[c] = np.linalg.norm(p - cen, axis=1).max() + R
    # anchor = the pose that minimises the worst displacement to the chunk's other poses
    best = None
    for a in range(k0, k1):
        dx = np.hypot(p[:, 0] - pose[a, 0], p[:, 1] - pose[a, 1]).max()
        if best is None or dx < best[0]:
            best = (dx, a)
    ka[c] = best[1]
    adx[c] = best[0]
    asm[c] = (2 * np.abs(np.sin(0.5 * (pose[k0:k1, 2] - pose[ka[c], 2])))).max()
sdf, ts, _ = o.query(w["points"], nthreads=os.cpu_count())
interior = w["points"][sdf < 0][:npts]
rng = np.random.default_rng(1)
print(f"{cfg}: K {K}, chunk {CH}, chunks {nch}, interior points used {len(interior)}; mean anchor slack dx {adx.mean():.3f} m, sin {asm.mean():.4f}")
L = 8   # lanes per sample scan
for r in (10.0, 6.0, 3.0, 1.5, 0.7, 0.3):
    st0 = st1 = st2 = st3 = 0
    ev0 = ev1 = ev2 = 0
    tot_cand = 0
    nq = 0
    for p in interior:
        th0 = rng.uniform(0, 2 * np.pi)
        ths = th0 + np.arange(21) * 0.3
        q = np.column_stack([p[0] + r * np.cos(ths), p[1] + r * np.sin(ths)])
        dpc = np.hypot(p[0] - ccx, p[1] - ccy)
        U = (dpc + crb).min() + r
        cand = np.nonzero(dpc - r - crb <= U)[0]
        if len(cand) > 48: cand = np.arange(nch)
        tot_cand += len(cand)
        for qq in q:
            dx, dy = qq[0] - pose[:, 0], qq[1] - pose[:, 1]
            rel = np.column_stack([cs * dx + sn * dy, -sn * dx + cs * dy])
            val = o.shape_eval(rel)                                     # all table values of this query
            cmin = np.array([val[CH * c:CH * c + CH].min() for c in range(nch)])
            truth = val.min()
            d = np.hypot(qq[0] - ccx[cand], qq[1] - ccy[cand])
            lb = d - crb[cand]
            alb = val[ka[cand]] - (adx[cand] + asm[cand] * np.hypot(dx[ka[cand]], dy[ka[cand]])) * (1 + 1e-9) - 1e-9
            assert np.all(alb <= cmin[cand] + 1e-12)
            # V0 today: nearest centre first, then list order with the running minimum (circle test)
            first = cand[np.argmin(d)]
            best = cmin[first]; n0 = 1
            for c, l in zip(cand, lb):
                if c == first: continue
                if l <= best:
                    n0 += 1; best = min(best, cmin[c])
            assert best == truth
            st0 += n0 * max(1, CH // L); ev0 += n0 * CH
            # V1: nearest chunk in full; circle survivors get their anchor (8 per step); then full steps where both bounds reach
            best = cmin[first]
            surv = [(c, i) for i, c in enumerate(cand) if c != first and lb[i] <= best]
            n_anchor_steps = (len(surv) + L - 1) // L
            if surv:
                best = min(best, min(val[ka[c]] for c, _ in surv))
            n1 = 0
            for c, i in surv:
                if lb[i] <= best and alb[i] <= best:
                    n1 += 1; best = min(best, cmin[c])
            assert best == truth
            st1 += max(1, CH // L) + n_anchor_steps + n1 * max(1, CH // L); ev1 += CH + len(surv) + n1 * CH
            # V2: anchors of ALL list entries first, then full steps (list order) where both bounds reach
            best = val[ka[cand]].min()
            n2 = 0
            for i, c in enumerate(cand):
                if lb[i] <= best and alb[i] <= best:
                    n2 += 1; best = min(best, cmin[c])
            assert best == truth
            st2 += (len(cand) + L - 1) // L + n2 * max(1, CH // L); ev2 += len(cand) + n2 * CH
            # V3: like V2, full steps in ascending order of the anchored bound
            best = val[ka[cand]].min()
            n3 = 0
            for i in np.argsort(np.maximum(alb, lb)):
                if max(lb[i], alb[i]) > best: break
                n3 += 1; best = min(best, cmin[cand[i]])
            assert best == truth
            st3 += (len(cand) + L - 1) // L + n3 * max(1, CH // L)
            nq += 1
    print(f"  r = {r:5.1f}: list {tot_cand / len(interior):5.1f}  steps/sample: today {st0 / nq:5.2f}  V1 near+anchor(circle survivors) {st1 / nq:5.2f}  "
          f"V2 anchors-all {st2 / nq:5.2f}  V3 anchors-all best-first {st3 / nq:5.2f}   table evals: {ev0 / nq:5.1f} / {ev1 / nq:5.1f} / {ev2 / nq:5.1f}")
