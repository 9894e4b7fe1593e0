"""Per-point bit comparison of one fuzz case against the device-arithmetic oracle (see tools/fuzz_parity.py):
usage: fuzz_case_bits.py <case> <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
from oracle import orc
case = int(sys.argv[1]); seed0 = int(sys.argv[2])
src = open(os.path.join(ROOT, "tools", "fuzz_parity.py")).read()
body = src.split("for case in range(ncase):\n", 1)[1].split("    ctx = svsdf_amd.SvsdfContext", 1)[0]
ns = dict(np=np, svsdf_amd=svsdf_amd, orc=orc, case=case, seed0=seed0, os=os, workload=workload)
exec("\n".join(l[4:] for l in body.split("\n")), ns)
shape, pp, poly, N, T, coeffs, pts, kw = (ns[k] for k in ("shape", "pp", "poly", "N", "T", "coeffs", "pts", "kw"))
print("shape", shape, "pp", pp, "N", N, "T", T)
ctx = svsdf_amd.SvsdfContext(shape=shape, device=0, **kw); ctx.set_points(pts)
o = orc.Oracle(shape, **kw); o.set_traj(coeffs, T); o.set_modes(1, 0)
sdf, tstar, g, _ = ctx.query_points(coeffs, T)
osdf, ots, og = o.query(pts, nthreads=os.cpu_count())
dt = tstar != ots; ds = sdf != osdf
print("t* differ", int(dt.sum()), "sdf differ", int(ds.sum()), "interior", int((osdf < 0).sum()), "interior among sdf-differ", int((osdf[ds] < 0).sum()),
      "exterior among sdf-differ", int((osdf[ds] >= 0).sum()))
ulp = lambda a, b: np.abs(a.view(np.int64) - b.view(np.int64))
print("sdf ulp distance: max", int(ulp(sdf[ds], osdf[ds]).max()), "median", int(np.median(ulp(sdf[ds], osdf[ds]))))
if dt.any():
    print("t* abs diff: max", np.abs(tstar[dt] - ots[dt]).max(), "median", np.median(np.abs(tstar[dt] - ots[dt])))
for i in np.where(ds)[0][:8]:
    print(f" pt {i} sdf hip {sdf[i]!r} orc {osdf[i]!r} t* hip {tstar[i]!r} orc {ots[i]!r}")
    if osdf[i] >= 0 and tstar[i] == ots[i]:
        print("    oracle sdf_at_time(t*):", repr(o.sdf_at_time(pts[i, 0], pts[i, 1], ots[i])))
