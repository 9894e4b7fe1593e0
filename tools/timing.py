import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
os.environ["SVSDF_LIB_VARIANT"] = "timing"
import numpy as np, svsdf_amd
from svsdf_amd import workload
w = workload.make(sys.argv[1] if len(sys.argv) > 1 else "C2", P=int(sys.argv[2]) if len(sys.argv) > 2 else None, minco=svsdf_amd.minco_coeffs)
ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                             poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
ctx.set_points(w["points"])
for i in range(3): ctx.eval_penalty(w["coeffs"], w["T"])
os.environ["SVSDF_DUMP_TIMING"] = "1"
ctx.eval_penalty(w["coeffs"], w["T"])
print(ctx.stats())
