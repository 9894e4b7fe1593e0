"""Diagnose one fuzz case (see tools/fuzz_parity.py): where do HIP and oracle disagree per point?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from oracle import orc
import importlib.util
case = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
src = open(os.path.join(ROOT, "tools", "fuzz_parity.py")).read()
# re-create the case by executing the generator part of the fuzz loop body
body = src.split("for case in range(ncase):\n", 1)[1].split("    ctx = svsdf_amd.SvsdfContext", 1)[0]
ns = dict(np=np, svsdf_amd=svsdf_amd, orc=orc, case=case, seed0=seed0)
exec("\n".join(l[4:] for l in body.split("\n")), ns)
shape, pp, poly, N, T, q, hs, ts, coeffs, pts, sh, kw = (ns[k] for k in ("shape", "pp", "poly", "N", "T", "q", "hs", "ts", "coeffs", "pts", "sh", "kw"))
print("shape", shape, "N", N, "T", np.round(T, 3), "kind", ns["kind"], "hs", hs[:, 0], "ts", ts[:, 0])
ctx = svsdf_amd.SvsdfContext(shape=shape, device=0, **kw); ctx.set_points(pts)
o = orc.Oracle(shape, **kw); o.set_traj(coeffs, T)
sdf, tstar, g, _ = ctx.query_points(coeffs, T)
osdf, ots, og = o.query(pts, nthreads=os.cpu_count())
fl = np.where(np.abs(tstar - ots) > 1e-6)[0]
print("flips", len(fl), "of", len(pts), "interior among flips", int((osdf[fl] <= 0).sum()))
for i in fl[:12]:
    print(f" pt {i} p=({pts[i,0]:.3f},{pts[i,1]:.3f}) sdf hip {sdf[i]:.12f} orc {osdf[i]:.12f} d={sdf[i]-osdf[i]:.2e}  t* hip {tstar[i]:.6f} orc {ots[i]:.6f}  g hip {np.round(g[i],4)} orc {np.round(og[i],4)}")
    if osdf[i] > 0:
        print("     oracle sdf_at_time at hip t*:", o.sdf_at_time(pts[i,0], pts[i,1], tstar[i]), "at orc t*:", o.sdf_at_time(pts[i,0], pts[i,1], ots[i]))
