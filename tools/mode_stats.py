"""Work counters of the GSIP bound modes on one workload: python tools/mode_stats.py NS 1000000"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
cfg, P = sys.argv[1], int(sys.argv[2])
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
for mode in sys.argv[3:] or ["0", "1", "2"]:
    os.environ["SVSDF_UB_FULL"] = mode
    c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                               poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    c.set_points(w["points"])
    for _ in range(3):
        out = c.eval_penalty(w["coeffs"], w["T"])
    t0 = time.perf_counter()
    for _ in range(5):
        out = c.eval_penalty(w["coeffs"], w["T"])
    ms = (time.perf_counter() - t0) / 5 * 1e3
    st = c.stats()
    c.set_profiling(True); c.eval_penalty(w["coeffs"], w["T"]); sp = c.stats(); c.set_profiling(False)
    print(mode, "ms %.2f" % ms, "solves/pt %.3f" % (st["solves"] / P), "samples/pt %.2f" % (st["gsip_samples"] / P),
          "scan evals/pt %.1f" % (st["scan_evals"] / P), "evals/pt %.1f" % (st["sdf_evals"] / P), "iters", st["gsip_iterations"],
          "solve_ms %.2f (sum %.2f) dev %.2f" % (sp["solve_ms"], sp["solve_ms_sum"], sp["device_ms"]), "cost", repr(out[0]))
    c.close()
