"""A few evaluations of one config for profiling runs: python tools/c5_driver.py [config] [points] [evaluations]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
import svsdf_amd
from svsdf_amd import workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                           poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                           tail_state=w["tail_state"], device=0)
c.set_points(w["points"])
for _ in range(n):
    out = c.eval_penalty(w["coeffs"], w["T"])
print(cfg, P, n, out[0], c.get_plan() if hasattr(c, "get_plan") else "")
