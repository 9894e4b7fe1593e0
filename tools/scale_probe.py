"""Main-solve latency/throughput probe: one evaluation per P (GPU); read durations from rocprof trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
import numpy as np, svsdf_amd
from svsdf_amd import workload
os.environ["SVSDF_BATCHES"] = "1"
for P in [16, 1024, 16384, 65536, 131072, 262144, 524288]:
    w = workload.make("C2", P=P, minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    ctx.eval_penalty(w["coeffs"], w["T"]); ctx.eval_penalty(w["coeffs"], w["T"])
    print("P", P, ctx.stats(), flush=True)
    ctx.close()
