"""One-time setup cost per optimisation (svsdf_set_points: host Morton sort + PCIe upload) vs one evaluation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
for cfg, P in (("C2", 100000), ("C3", 1000000)):
    w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    pts = np.ascontiguousarray(w["points"])
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.set_points(pts); ts.append(time.perf_counter() - t0)
    ev = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.eval_penalty(w["coeffs"], w["T"]); ev.append(time.perf_counter() - t0)
    print(f"{cfg} P={P}: set_points {min(ts)*1e3:.2f} ms, evaluation {min(ev)*1e3:.2f} ms, "
          f"upload-every-evaluation rate {P/(min(ts)+min(ev))/1e6:.2f} M points/s vs resident {P/min(ev)/1e6:.2f} M points/s")
