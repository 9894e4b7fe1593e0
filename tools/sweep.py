"""Timing sweep over env-var variants (GPU)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
import numpy as np, svsdf_amd
from svsdf_amd import workload
cfg = sys.argv[1]; P = int(sys.argv[2])
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
ref = None
for spec in sys.argv[3:]:
    env = dict(kv.split("=") for kv in spec.split(","))
    for k in [k for k in os.environ if k.startswith("SVSDF_") and k != "SVSDF_LIB_VARIANT"]:
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); out = ctx.eval_penalty(w["coeffs"], w["T"]); ts.append(time.perf_counter() - t0)
    st = ctx.stats(); ctx.close()
    if ref is None: ref = out
    same = out[0] == ref[0] and np.array_equal(out[2], ref[2])
    print(f"{spec:60s} best {min(ts)*1e3:7.3f} ms med {np.median(ts)*1e3:7.3f}  refine_sum={st['solve_ms']:.2f} dev={st['device_ms']:.2f} same={same}")
