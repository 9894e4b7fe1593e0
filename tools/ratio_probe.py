"""Round 5: the counters the GSIP bound-mode rule looks at after the FIRST evaluation of a point set (interior points, GSIP samples,
solves, bound_ratio = GSIP solves / samples) on small clouds of every BASELINE config.  usage: ratio_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
def run(w, tag):
    c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                               poly_params=w["poly_params"], polygon=w.get("polygon"), head_state=w["head_state"],
                               tail_state=w["tail_state"], device=0)
    c.set_points(w["points"])
    c.eval_penalty(w["coeffs"], w["T"])
    st = c.stats()
    print(tag, "P", len(w["points"]), "interior", st["interior_points"], "bound_ratio %.3f" % st["bound_ratio"], "samples", st["gsip_samples"], "solves", st["solves"], flush=True)
    c.close()
for cfg in ("C1", "C2", "NS", "C3", "C4"):
    for P in (300, 1000, 3000, 10000):
        run(workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs), cfg)
