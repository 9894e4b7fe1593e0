"""One-off parity run at BASELINE.json's FULL sizes (GPU box, 256 host cores; minutes of CPU): HIP path vs the oracle
of record on every point of the workload.  usage: python tools/full_size_parity.py NS C3 C4 C5:100000
(CFG:P limits a workload to its first P points: the 77-vertex Polygon of C5 costs the oracle ~1.2 ms per point and core)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
from oracle import orc
NT = os.cpu_count() or 1
rel = lambda a, b: float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))
for arg in sys.argv[1:]:
    cfg, _, lim = arg.partition(":")
    w = workload.make(cfg, P=int(lim) if lim else None, minco=svsdf_amd.minco_coeffs)
    P = len(w["points"])
    kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], poly_params=w["poly_params"],
              polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"])
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], device=0, **kw)
    ctx.set_points(w["points"])
    t0 = time.time()
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    cost, gT, gC = ctx.eval_penalty(w["coeffs"], w["T"])
    st = ctx.stats()
    t_hip = time.time() - t0
    o = orc.Oracle(w["shape"], **kw)
    o.set_traj(w["coeffs"], w["T"])
    t0 = time.time()
    ocost, ogT, ogC, osdf, ots, _ = o.penalty(w["points"], nthreads=NT, sum_mode=1, per_point=True)
    t_orc = time.time() - t0
    flips = np.abs(ts - ots) > 1e-6
    ok = ~flips
    print(f"{cfg}: P = {P}, basin flips {int(flips.sum())} ({flips.mean():.2e}), max |dsdf| (no flip) {np.abs(sdf[ok] - osdf[ok]).max():.2e}, "
          f"interior {st['interior_points']} vs {o.counters()['interior_points']}, cost rel {abs(cost - ocost) / abs(ocost):.2e}, "
          f"gradC rel {rel(gC, ogC):.2e}, gradT rel {rel(gT, ogT):.2e}; HIP {t_hip:.2f} s, oracle {t_orc:.1f} s on {NT} threads", flush=True)
    ctx.close()
