"""Driver for a rocprofv3 kernel trace at the reference's scale: one demo map through the producer, 24 pieces, a few full
callbacks (python tools/ref_trace.py star [calls])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import svsdf_amd
from svsdf_amd import workload
name = sys.argv[1] if len(sys.argv) > 1 else "star"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w = workload.reference_case(name, N=24)
c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                           poly_params=w["poly_params"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
c.set_points(w["points"])
for k in range(n):
    c.lmbm_evaluate(w["xs"][k % len(w["xs"])])
print(len(w["points"]), c.stats())
