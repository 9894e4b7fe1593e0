"""Exact-cull check (GPU): eval_penalty with SVSDF_CULL=1 vs 0, and every culled point is inactive by query_points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"; P = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
res = {}
for cull in ("1", "0"):
    os.environ["SVSDF_CULL"] = cull
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); out = ctx.eval_penalty(w["coeffs"], w["T"]); ts.append(time.perf_counter() - t0)
    st = ctx.stats()
    res[cull] = out
    print("cull", cull, "best ms", min(ts) * 1e3, "culled", st["culled_points"], "solves", st["solves"], "cost", out[0])
    if cull == "1":
        sdf, tstar, g, _ = ctx.query_points(w["coeffs"], w["T"])
        print("   inactive by true sdf:", int((sdf > w["safety_hor"]).sum()), "of", len(sdf))
a, b = res["1"], res["0"]
print("rel diff cost", abs(a[0] - b[0]) / abs(b[0]), "gT", np.abs(a[1] - b[1]).max() / np.abs(b[1]).max(), "gC", np.abs(a[2] - b[2]).max() / np.abs(b[2]).max())
