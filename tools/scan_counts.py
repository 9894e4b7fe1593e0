"""k_round table evaluations and GSIP solve counts of one workload (round 4: anchor scans).  usage: scan_counts.py <variant|-> cfg P"""
import os, sys
if sys.argv[1] != "-": os.environ["SVSDF_LIB_VARIANT"] = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import svsdf_amd
from svsdf_amd import workload
cfg, P = sys.argv[2], int(sys.argv[3])
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], poly_params=w["poly_params"],
                           polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
c.set_points(w["points"])
for _ in range(4): c.eval_penalty(w["coeffs"], w["T"])
c.set_profiling(2); c.eval_penalty(w["coeffs"], w["T"]); c.eval_penalty(w["coeffs"], w["T"])
st = c.stats()
print(sys.argv[1], cfg, P, "mode", st["gsip_bound_mode"], "samples", st["gsip_samples"], "round table evals", st["round_scan_evals"], "per sample %.1f" % (st["round_scan_evals"] / max(st["gsip_samples"], 1)),
      "solves", st["solves"], "k_round ms (one batch) %.3f" % st["round_ms"], "k_solve ms %.3f" % st["solve_ms"], "device ms %.3f" % st["device_ms"])
