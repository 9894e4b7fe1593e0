"""Where do the device and the device-arithmetic oracle part ways in one fuzz case (tools/fuzz_parity.py)?  SDF-at-time of the
case's points at the oracle's t*, through svsdf_debug_sdf_at, against the oracle's pose, transform and shape value.
usage: fuzz_case_parts.py <case> <seed>"""
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
ctx = svsdf_amd.SvsdfContext(shape=shape, device=0, **kw); ctx.set_points(pts)
o = orc.Oracle(shape, **kw); o.set_traj(coeffs, T); o.set_modes(1, 0)
osdf, ots, og = o.query(pts, nthreads=os.cpu_count())
ext = osdf > 0
xy = pts[ext, :2]; tt = ots[ext]
d = ctx.debug_sdf_at(coeffs, T, xy, tt)
oval = np.array([o.sdf_at_time(x, y, t) for (x, y), t in zip(xy, tt)])
opos = np.array([o.pos(t) for t in tt])
print("exterior points", len(tt), " sdf differ", int((d[:, 0] != oval).sum()), " pose x differ", int((d[:, 1] != opos[:, 0]).sum()),
      " pose y differ", int((d[:, 2] != opos[:, 1]).sum()), " piece-time mode", d[0, 7])
dx, dy = xy[:, 0] - d[:, 1], xy[:, 1] - d[:, 2]
rx = d[:, 3] * dx + d[:, 4] * dy
ry = (-d[:, 4]) * dx + d[:, 3] * dy
print("transform (device cos/sin, numpy arithmetic) differs from the device's body-frame point:", int((rx != d[:, 5]).sum()), int((ry != d[:, 6]).sum()))
sv = o.shape_eval(np.column_stack([d[:, 5], d[:, 6]]))
print("oracle shape value at the DEVICE's body-frame point differs from the device's sdf:", int((sv != d[:, 0]).sum()))
cs, sn = np.cos(opos[:, 2]), np.sin(opos[:, 2])
print("device cos / sin differ from numpy's (libm) at the oracle's yaw:", int((cs != d[:, 3]).sum()), int((sn != d[:, 4]).sum()))
bad = np.where(sv != d[:, 0])[0][:6]
for i in bad:
    print(f"  body point ({d[i,5]!r}, {d[i,6]!r}): device {d[i,0]!r} oracle shape {sv[i]!r}")
