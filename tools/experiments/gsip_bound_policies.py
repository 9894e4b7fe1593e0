"""CPU study (round 6): what GSIP bound policy costs how many table evaluations and solves.

Replays the reference's GSIP rounds (SWM:926-1017) for interior points of a workload with the oracle's own solve as the
sample value, and for every round records, per sample, the exact table minimum T_j and several cheap upper bounds:
  near   the 8 poses of the chunk with the nearest centre                      (today's cheap bound: 8 evaluations)
  anch8  the anchor pose (middle) of the 8 chunks with the nearest centres      (8 evaluations)
  both   min(near, anch8)                                                       (16 evaluations)
  hint   the 8 poses of the chunk that holds the point's current t* (res_t)    (8 evaluations)
Policies are then priced in table evaluations + solves per point-round:
  full   scan every sample, request the samples within delta of the best bound
  lazy(B) bound B for all; scan the band [max B - band, ..]; extend while unscanned bounds reach best scanned - delta
usage: python tools/experiments/gsip_bound_policies.py [config] [n_points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np
from svsdf_amd import workload
from oracle import orc

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 60
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
nch = (K + 7) // 8
ccx = np.array([0.5 * (pose[8 * c:8 * c + 8, 0].min() + pose[8 * c:8 * c + 8, 0].max()) for c in range(nch)])
ccy = np.array([0.5 * (pose[8 * c:8 * c + 8, 1].min() + pose[8 * c:8 * c + 8, 1].max()) for c in range(nch)])
ka = np.array([min(8 * c + 3, K - 1) for c in range(nch)])
RB = {"sdHorseshoe": float(np.hypot(1.7, 1.55)), "star": 2.8, "sdHeart": 4.0 * (np.sqrt(0.25 ** 2 + 0.75 ** 2) + np.sqrt(2) / 4)}
crb = np.array([np.hypot(pose[8 * c:8 * c + 8, 0] - ccx[c], pose[8 * c:8 * c + 8, 1] - ccy[c]).max() for c in range(nch)]) + RB[w["shape"]] + 1e-6
sdf, ts, _ = o.query(w["points"], nthreads=os.cpu_count())
sel = np.nonzero(sdf < 0)[0][:npts]
PI = 3.14159265358979323846
DELTA, BAND = 0.01, 0.1

def table(q):
    dx, dy = q[0] - pose[:, 0], q[1] - pose[:, 1]
    return o.shape_eval(np.column_stack([cs * dx + sn * dy, -sn * dx + cs * dy]))

rounds = []   # per point-round: dict(r, T[], g[], bounds{name: []})
for i in sel:
    p = w["points"][i, :2]
    t_star = ts[i]
    v = o.vel(t_star)
    th0 = np.arctan2(v[0], -v[1])
    if th0 < 0: th0 += 2 * PI
    r, thres, it = 10.0, PI + 0.1, 1
    while True:
        ths = []
        th = th0
        while th < th0 + 2 * PI and len(ths) < 24:
            ths.append(th); th += thres
        T, g, tt, B = [], [], [], {"near": [], "anch8": [], "hint": [], "alist": [], "alist+near": []}
        # candidate list of the round (round_point)
        dpc = np.hypot(p[0] - ccx, p[1] - ccy)
        U = (dpc + crb).min() + abs(r)
        cand = np.nonzero(dpc - abs(r) - crb <= U)[0]
        has_list = len(cand) <= 48
        ncost = []
        hint_c = min(int(round(t_star / 0.15)) // 8, nch - 1)
        for th in ths:
            q = (p[0] + r * np.cos(th), p[1] + r * np.sin(th))
            val = table(q)
            T.append(val.min())
            gv, gt, _ = o.sdf_swept(q[0], q[1])
            g.append(gv); tt.append(gt)
            d2 = (q[0] - ccx) ** 2 + (q[1] - ccy) ** 2
            order = np.argsort(d2)
            c0 = order[0]
            B["near"].append(val[8 * c0:8 * c0 + 8].min())
            B["anch8"].append(val[ka[order[:8]]].min())
            B["hint"].append(val[8 * hint_c:8 * hint_c + 8].min())
            B["alist"].append(val[ka[cand]].min() if has_list else B["near"][-1])
            B["alist+near"].append(min(val[ka[cand]].min(), B["near"][-1]) if has_list else B["near"][-1])
            ncost.append(len(cand) if has_list else 8)
        B["both"] = list(np.minimum(B["near"], B["anch8"]))
        B["near+hint"] = list(np.minimum(B["near"], B["hint"]))
        rounds.append(dict(r=r, it=it, T=np.array(T), g=np.array(g), B={k: np.array(v_) for k, v_ in B.items()}, ncost=np.array(ncost), nl=len(cand)))
        jm = int(np.argmax(g)); mg = g[jm]
        r_star = r - mg
        if it > 8 or abs(mg) < 0.1: break
        thres = max(0.3, thres / 3); r = r_star; th0 = ths[jm]; t_star = tt[jm]; it += 1

print(f"{cfg}: {len(sel)} interior points, {len(rounds)} point-rounds, samples per round {np.mean([len(x['T']) for x in rounds]):.1f}")
SCAN = 43.7   # table evaluations of one full scan (measured mean)
def price(policy, bname=None, cost_b=0):
    ev = sv = ns = 0
    for R in rounds:
        T, g = R["T"], R["g"]; n = len(T)
        if policy == "full":
            ub = T.copy(); scanned = np.ones(n, bool); ev += n * SCAN
        else:
            Bv = R["B"][bname]; ev += (R["ncost"].sum() + (8 * n if bname == "alist+near" and R["nl"] <= 48 else 0)) if bname.startswith("alist") else n * cost_b
            scanned = Bv >= Bv.max() - BAND
            ub = np.where(scanned, T, Bv)
            for rep in range(3):
                u2 = ub[scanned].max()
                ext = (~scanned) & (Bv >= u2 - DELTA)
                if not ext.any(): break
                scanned |= ext; ub = np.where(scanned, T, Bv)
            ev += scanned.sum() * SCAN
        ns += scanned.sum()
        req = scanned & (ub >= ub[scanned].max() - DELTA)
        gstar = g[req].max()
        supp = (~req) & (ub >= gstar)      # closing the round: unsolved samples whose bound still reaches the best solved value
        while supp.any():
            req |= supp; gstar = g[req].max(); supp = (~req) & (ub >= gstar)
        sv += req.sum()
    n_r = len(rounds)
    return ev / n_r, ns / n_r, sv / n_r
print(f"{'policy':16s} table evals / point-round   scans / point-round   solves / point-round   (a solve ~ 250 full evaluations ~ 680 table-evaluation equivalents)")
for name, args in [("full", ("full",)), ("lazy(near)", ("lazy", "near", 8)), ("lazy(anch8)", ("lazy", "anch8", 8)), ("lazy(both)", ("lazy", "both", 16)),
                   ("lazy(hint)", ("lazy", "hint", 8)), ("lazy(near+hint)", ("lazy", "near+hint", 16)), ("lazy(alist)", ("lazy", "alist", 0)), ("lazy(alist+near)", ("lazy", "alist+near", 0))]:
    ev, ns, sv = price(*args)
    print(f"{name:16s} {ev:10.1f} {ns:22.2f} {sv:20.2f}      total ~ {ev + 680 * sv:8.0f}")
for rr in (1, 2, 3, 4, 5, 6):
    sub = [R for R in rounds if R["it"] == rr]
    if sub: print(f"  round {rr}: {len(sub)} point-rounds, mean r {np.mean([abs(R['r']) for R in sub]):.2f}, samples {np.mean([len(R['T']) for R in sub]):.1f}, "
                  f"looseness near {np.mean([np.mean(R['B']['near'] - R['T']) for R in sub]):.3f} anch8 {np.mean([np.mean(R['B']['anch8'] - R['T']) for R in sub]):.3f} both {np.mean([np.mean(R['B']['both'] - R['T']) for R in sub]):.3f} hint {np.mean([np.mean(R['B']['hint'] - R['T']) for R in sub]):.3f} alist {np.mean([np.mean(R['B']['alist'] - R['T']) for R in sub]):.3f} (list {np.mean([R['nl'] for R in sub]):.1f})")
