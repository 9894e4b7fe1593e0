import os, sys
ROOT = "/root/repo"
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
for name in ("star", "sdHorseshoe", "sdHeart"):
    rc = workload.reference_case(name)
    w = rc["workload"] if "workload" in rc else rc
    try:
        run(dict(w, coeffs=rc["iterates"][0]["coeffs"], T=rc["iterates"][0]["T"]) if "iterates" in rc else w, "ref:" + name)
    except Exception as ex:
        print("ref", name, "skipped:", ex, list(rc.keys())[:12])
