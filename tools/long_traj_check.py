"""One-off (round 5): how long a trajectory the LDS pose table takes.  N pieces of `piece_s` seconds, 3 000 points, penalty against
the oracle.  usage: long_traj_check.py N piece_s [N piece_s ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
from oracle import orc
args = sys.argv[1:]
for i in range(0, len(args), 2):
    N, ps = int(args[i]), float(args[i + 1])
    w = workload.make("C3", P=3000, N=N)
    w["T"] = ps * (1.0 + 1e-3 * np.random.default_rng(N).standard_normal(N))
    w["coeffs"] = svsdf_amd.minco_coeffs(w["head_state"], w["tail_state"], w["q"], w["T"])
    kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], poly_params=w["poly_params"],
              head_state=w["head_state"], tail_state=w["tail_state"])
    c = svsdf_amd.SvsdfContext(shape=w["shape"], device=0, **kw)
    c.set_points(w["points"])
    try:
        for _ in range(3):
            cost, gT, gC = c.eval_penalty(w["coeffs"], w["T"])
        t0 = time.perf_counter()
        for _ in range(10):
            c.eval_penalty(w["coeffs"], w["T"])
        ms = 1e2 * (time.perf_counter() - t0)
    except Exception as ex:
        print(f"N {N} x {ps} s = {w['T'].sum():.0f} s: refused: {ex}", flush=True)
        continue
    o = orc.Oracle(w["shape"], **kw)
    o.set_traj(w["coeffs"], w["T"])
    oc, ogT, ogC = o.penalty(w["points"], nthreads=os.cpu_count(), sum_mode=1)
    rel = lambda a, b: float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))
    print(f"N {N} x {ps} s = {w['T'].sum():.0f} s: {ms:.3f} ms per evaluation, cost rel {abs(cost - oc) / abs(oc):.2e}, gradC rel {rel(gC, ogC):.2e}, "
          f"gradT rel {rel(gT, ogT):.2e}, plan {c.get_plan()}", flush=True)
    c.close()
