"""Round 5: where do the rare slow evaluations come from?  N back-to-back evaluations (generic durations), per-step wall time;
prints every step slower than 1.5 x the median with its index, the launches it issued and the gap to the previous slow one.
usage: stall_probe.py <config> <points> <steps> [env SVSDF_* as usual]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
cfg, P, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if os.environ.get("PROBE_NO_GC"):
    gc.disable()
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
N = len(w["T"])
T = w["T"] * (1.0 + 1e-3 * np.random.default_rng(11).standard_normal(N))
coeffs = svsdf_amd.minco_coeffs(w["head_state"], w["tail_state"], w["q"], T)
c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                           poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                           tail_state=w["tail_state"], device=0)
c.set_points(w["points"])
for _ in range(8):
    c.eval_penalty(coeffs, T)
per = np.zeros(steps)
for k in range(steps):
    t0 = time.perf_counter()
    c.eval_penalty(coeffs, T)
    per[k] = 1e3 * (time.perf_counter() - t0)
med = float(np.median(per))
slow = np.nonzero(per > 1.5 * med)[0]
st = c.stats()
print(f"{cfg} P={P}: {steps} steps, median {med:.3f} ms, mean {per.mean():.3f} ms, launches per step {st['solve_launches']} (solve) , "
      f"{len(slow)} slow steps = {100.0 * (per[slow] - med).sum() / per.sum():.2f} % of the total time")
prev = None
for i in slow:
    print(f"  step {int(i):6d}  {per[i]:8.3f} ms  gap {'' if prev is None else int(i - prev)}")
    prev = i
