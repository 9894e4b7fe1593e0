"""Minimal driver for rocprofv3: set_points + a few evaluations of one workload (python tools/prof_eval.py C3 1000000 [evals])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import svsdf_amd
from svsdf_amd import workload
cfg, P = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                           poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                           tail_state=w["tail_state"], device=0)
c.set_points(w["points"])
for _ in range(n):
    c.eval_penalty(w["coeffs"], w["T"])
print(c.stats())
