import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
import numpy as np, svsdf_amd
from svsdf_amd import workload
w = workload.make("C2", minco=svsdf_amd.minco_coeffs)
ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                             head_state=w["head_state"], tail_state=w["tail_state"], device=0)
ctx.set_points(w["points"])
for i in range(12):
    t0 = time.perf_counter(); ctx.eval_penalty(w["coeffs"], w["T"]); dt = time.perf_counter() - t0
    st = ctx.stats()
    print(i, f"{dt*1e3:.3f} ms", st["gsip_iterations"], st["solve_launches"], st["solves"], flush=True)
