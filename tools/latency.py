"""Evaluation latency vs shard size (GPU): the reference's demo maps give ~10^2 points, C1 10^4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
for P in (85, 1000, 10000, 30000, 100000):
    w = workload.make("C1" if P <= 10000 else "C2", P=P, minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); ctx.eval_penalty(w["coeffs"], w["T"]); ts.append(time.perf_counter() - t0)
    st = ctx.stats()
    print(f"P={P:6d} N={len(w['T'])}: best {min(ts)*1e3:.3f} ms  median {np.median(ts)*1e3:.3f} ms  launches {st['solve_launches']} iterations {st['gsip_iterations']} interior {st['interior_points']}")
