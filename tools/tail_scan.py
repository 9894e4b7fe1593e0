"""k_tail duration against the number of GSIP points it holds (round 4): C1 workload at growing point counts, the whole GSIP
loop in the tail (svsdf_set_plan tail_iter = 0) against the launch chain (tail_iter = -2).  usage: tail_scan.py [variant]"""
import os, sys, time
if len(sys.argv) > 1 and sys.argv[1] != "-":
    os.environ["SVSDF_LIB_VARIANT"] = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
for P in (3000, 6000, 9000, 10000, 11000, 12000, 13000, 14000, 15000, 20000, 30000):
    w = workload.make("C1", P=P, minco=svsdf_amd.minco_coeffs)
    row = [P]
    for ti in (-2, 0):
        c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                   head_state=w["head_state"], tail_state=w["tail_state"], device=0)
        c.set_points(w["points"])
        c.set_plan(tail_iter=ti)
        for _ in range(4):
            c.eval_penalty(w["coeffs"], w["T"])
        t0 = time.perf_counter()
        for _ in range(20):
            c.eval_penalty(w["coeffs"], w["T"])
        ms = 1e3 * (time.perf_counter() - t0) / 20
        c.set_profiling(True)
        c.eval_penalty(w["coeffs"], w["T"])
        st = c.stats()
        row += [round(ms, 3), round(st["tail_ms"], 3), st["tail_points"], st["interior_points"]]
        c.close()
    print("P %6d  chain %.3f ms | tail %.3f ms (k_tail %.3f ms, %d points in it, %d interior)" % (row[0], row[1], row[5], row[6], row[7], row[8]), flush=True)
