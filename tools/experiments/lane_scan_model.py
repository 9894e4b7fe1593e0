"""CPU study (round 6): the lane-per-sample anchored seed scan of k_round (scan_sample_lane), modelled per wave.

Per sample (one lane): nearest chunk c0 in full (8 poses), circle test over the candidate list, the anchor pose of every
survivor (running minimum; anchored Lipschitz bound tested at once), then the surviving chunks in full with the circle
test repeated against the running minimum.  A wave executes the MAXIMUM trip count among its lanes: the model prints table
evaluations per sample and wave-instructions per sample for LP = 32 (two points of <= 21 samples per wave).
usage: python tools/experiments/lane_scan_model.py [config] [n_points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np
from svsdf_amd import workload
from oracle import orc

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 100
RB = {"sdHorseshoe": float(np.hypot(1.7, 1.55)), "star": 2.8, "sdHeart": 4.0 * (np.sqrt(0.25 ** 2 + 0.75 ** 2) + np.sqrt(2) / 4)}
w = workload.make(cfg, P=20000, minco=orc.minco_coeffs)
o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], head_state=w["head_state"], tail_state=w["tail_state"])
o.set_traj(w["coeffs"], w["T"])
dur = o.duration()
tk = []
t = 0.0
while t <= dur:
    tk.append(t); t += 0.15
tk = np.array(tk); K = len(tk)
pose = np.array([o.pos(t) for t in tk])
cs, sn = np.cos(pose[:, 2]), np.sin(pose[:, 2])
CH = 8
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
    ka[c] = min(k0 + 3, k1 - 1)
    adx[c] = np.hypot(p[:, 0] - pose[ka[c], 0], p[:, 1] - pose[ka[c], 1]).max()
    asm[c] = np.hypot(cs[k0:k1] - cs[ka[c]], sn[k0:k1] - sn[ka[c]]).max()
sdf, ts, _ = o.query(w["points"], nthreads=os.cpu_count())
interior = w["points"][sdf < 0][:npts]
rng = np.random.default_rng(1)
EV, ANCH = 95, 115          # wave-instructions: one table evaluation; one anchor (evaluation + bound)
print(f"{cfg}: K {K}, chunks {nch}, interior points {len(interior)}")
for r in (10.0, 6.0, 3.0, 1.5, 0.7, 0.3):
    per_point = []
    evals = 0; nq = 0
    for p in interior:
        th0 = rng.uniform(0, 2 * np.pi)
        ths = th0 + np.arange(21) * 0.3
        q = np.column_stack([p[0] + r * np.cos(ths), p[1] + r * np.sin(ths)])
        dpc = np.hypot(p[0] - ccx, p[1] - ccy)
        U = (dpc + crb).min() + r
        cand = np.nonzero(dpc - r - crb <= U)[0]
        if len(cand) > 48: cand = np.arange(nch)
        na = []; nf = []
        for qq in q:
            dx, dy = qq[0] - pose[:, 0], qq[1] - pose[:, 1]
            rel = np.column_stack([cs * dx + sn * dy, -sn * dx + cs * dy])
            val = o.shape_eval(rel)
            truth = val.min()
            d2 = (qq[0] - ccx[cand]) ** 2 + (qq[1] - ccy[cand]) ** 2
            c0 = cand[np.argmin(d2)]
            best = val[CH * c0:CH * c0 + CH].min()
            surv = [c for c, dd in zip(cand, d2) if c != c0 and best + crb[c] >= 0 and dd <= (best + crb[c]) ** 2 * (1 + 1e-12)]
            full = []
            for c in surv:
                v = val[ka[c]]
                best = min(best, v)
                dist = np.hypot(dx[ka[c]], dy[ka[c]]) * 1.0001
                alb = v - (adx[c] + asm[c] * dist) * (1 + 1e-9) - 1e-9
                if alb <= best: full.append(c)
            n_full = 0
            for c in full:
                dd = (qq[0] - ccx[c]) ** 2 + (qq[1] - ccy[c]) ** 2
                if best + crb[c] >= 0 and dd <= (best + crb[c]) ** 2 * (1 + 1e-12):
                    n_full += 1
                    best = min(best, val[CH * c:CH * c + CH].min())
            assert best == truth, (best, truth)
            na.append(len(surv)); nf.append(n_full)
            evals += 8 + len(surv) + 8 * n_full; nq += 1
        per_point.append((len(cand), max(na), max(nf), np.mean(na), np.mean(nf)))
    pp = np.array(per_point)
    # a wave = two points: trip counts = the maximum over both
    waves = [(max(pp[i, 0], pp[i + 1, 0]), max(pp[i, 1], pp[i + 1, 1]), max(pp[i, 2], pp[i + 1, 2])) for i in range(0, len(pp) - 1, 2)]
    cost = np.mean([2 * nl * 10 + 8 * EV + ma * ANCH + mf * 8 * EV for nl, ma, mf in waves])
    print(f"  r = {r:5.1f}: list {pp[:, 0].mean():5.1f}  anchors/sample {pp[:, 3].mean():4.2f} (max in a point {pp[:, 1].mean():4.2f})  full chunks/sample {pp[:, 4].mean():4.2f} (max {pp[:, 2].mean():4.2f})"
          f"  table evals/sample {evals / nq:5.1f}   wave-instr per wave {cost:6.0f} = {cost / 42:5.1f} per sample")
