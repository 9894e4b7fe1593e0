"""CPU study (round 6): seed-scan steps per GSIP sample -- today's chunk walk (circle test, 8 poses per step) against a
POSE list per (point, round) built from the table values at the round's CENTRE p:  every sample q of the circle |q - p| = |r|
has  sdf_k(p) - |r| <= sdf_k(q) <= sdf_k(p) + |r|  (1-Lipschitz shape SDF), so a pose with sdf_k(p) > min_k sdf_k(p) + 2 |r|
can never be, or tie with, any sample's table minimum.  The surviving poses (ascending k) are scanned 8 per step; a step whose
8 poses all have sdf_k(p) - |r| > running minimum is skipped.
usage: python tools/experiments/center_pose_list.py [config] [n_points]"""
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
RB = {"sdHorseshoe": float(np.hypot(1.7, 1.55)), "star": 2.8, "sdHeart": 4.0 * (np.sqrt(0.25 ** 2 + 0.75 ** 2) + np.sqrt(2) / 4)}
crb = np.array([np.hypot(pose[8 * c:8 * c + 8, 0] - ccx[c], pose[8 * c:8 * c + 8, 1] - ccy[c]).max() for c in range(nch)]) + RB[w["shape"]] + 1e-6
sdf, ts, _ = o.query(w["points"], nthreads=os.cpu_count())
sel = np.nonzero(sdf < 0)[0][:npts]
PI = 3.14159265358979323846

def table(q):
    dx, dy = q[0] - pose[:, 0], q[1] - pose[:, 1]
    return o.shape_eval(np.column_stack([cs * dx + sn * dy, -sn * dx + cs * dy]))

def today_steps(q, val, cand):
    d2 = (q[0] - ccx[cand]) ** 2 + (q[1] - ccy[cand]) ** 2
    c0 = cand[int(np.argmin(d2))]
    best = val[8 * c0:8 * c0 + 8].min(); steps = 1
    for c in cand:
        if c == c0: continue
        tt = best + crb[c]
        if tt >= 0 and (q[0] - ccx[c]) ** 2 + (q[1] - ccy[c]) ** 2 <= tt * tt:
            best = min(best, val[8 * c:8 * c + 8].min()); steps += 1
    return steps, best

stats = {}
for i in sel:
    p = w["points"][i, :2]
    t_star = ts[i]
    v = o.vel(t_star)
    th0 = np.arctan2(v[0], -v[1])
    if th0 < 0: th0 += 2 * PI
    r, thres, it = 10.0, PI + 0.1, 1
    Tc = table(p)
    while True:
        ths = []
        th = th0
        while th < th0 + 2 * PI and len(ths) < 24:
            ths.append(th); th += thres
        dpc = np.hypot(p[0] - ccx, p[1] - ccy)
        U = (dpc + crb).min() + abs(r)
        cand = np.nonzero(dpc - abs(r) - crb <= U)[0]
        if len(cand) > 48 or abs(r) >= 8: cand = np.arange(nch)
        # pose list from the centre values (only poses of the candidate chunks are evaluated at the centre)
        inlist = np.zeros(K, bool)
        for c in cand: inlist[8 * c:8 * c + 8] = True
        m = Tc[inlist].min()
        pl = np.nonzero(inlist & (Tc <= m + 2 * abs(r) + 1e-9))[0]
        # order variants: ascending k / ascending centre value
        pl_sorted = pl[np.argsort(Tc[pl], kind="stable")]
        g = []; tt_ = []
        st_today = st_new = st_new_sorted = 0
        for th in ths:
            q = (p[0] + r * np.cos(th), p[1] + r * np.sin(th))
            val = table(q)
            s0, b0 = today_steps(q, val, cand)
            st_today += s0
            for order, key in ((pl, "k"), (pl_sorted, "s")):
                best = 1e300; s1 = 0
                for j in range(0, len(order), 8):
                    grp = order[j:j + 8]
                    if (Tc[grp] - abs(r)).min() > best: continue
                    best = min(best, val[grp].min()); s1 += 1
                assert abs(best - val.min()) < 1e-12 or not np.array_equal(cand, np.arange(nch)) and abs(best - val[inlist].min()) < 1e-12, (best, val.min())
                if key == "k": st_new += s1
                else: st_new_sorted += s1
            gv, gt, _ = o.sdf_swept(q[0], q[1])
            g.append(gv); tt_.append(gt)
        n = len(ths)
        S = stats.setdefault(it, dict(n=0, r=0.0, samples=0, today=0, new=0, news=0, poses=0, listch=0))
        S["n"] += 1; S["r"] += abs(r); S["samples"] += n; S["today"] += st_today; S["new"] += st_new; S["news"] += st_new_sorted; S["poses"] += len(pl); S["listch"] += len(cand)
        jm = int(np.argmax(g)); mg = g[jm]
        r_star = r - mg
        if it > 8 or abs(mg) < 0.1: break
        thres = max(0.3, thres / 3); r = r_star; th0 = ths[jm]; t_star = tt_[jm]; it += 1

print(f"{cfg}: K {K}, chunks {nch}, interior points {len(sel)}")
tot = dict(samples=0, today=0, new=0, news=0, centre=0.0)
for it in sorted(stats):
    S = stats[it]; n = S["n"]
    centre = S["listch"] / n   # 8-lane steps to evaluate the centre's table over the list chunks
    print(f"  round {it}: {n:4d} point-rounds  mean |r| {S['r'] / n:6.2f}  samples {S['samples'] / n:5.1f}  list chunks {S['listch'] / n:5.1f}  pose list {S['poses'] / n:6.1f}"
          f"   8-lane steps per sample: today {S['today'] / S['samples']:.2f}  pose list {S['new'] / S['samples']:.2f} (sorted by centre value {S['news'] / S['samples']:.2f})"
          f"  + centre {centre:.1f} steps per round = {(S['new'] + centre * n) / S['samples']:.2f} per sample")
    for k in ("samples", "today", "new", "news"): tot[k] += S[k]
    tot["centre"] += centre * n
print(f"  all rounds: today {tot['today'] / tot['samples']:.2f} steps per sample; pose list {(tot['new'] + tot['centre']) / tot['samples']:.2f} (sorted {(tot['news'] + tot['centre']) / tot['samples']:.2f}) incl. the centre evaluation")
