"""CPU-only study (oracle): how many passes do the descents of one evaluation take?

usage: python tools/experiments/pass_stats.py [config] [points]
A pass of gradientDescent (SWM:1249-1325) = one FD derivative + one halving ladder, strictly one after the other; the
slowest descent of a launch bounds the launch from below whatever the throughput."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd"), os.path.join(ROOT, "oracle")]
import orc  # noqa: E402
from svsdf_amd import workload  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
w = workload.make(cfg, P=P, minco=orc.minco_coeffs)
o = orc.Oracle(w["shape"], poly_params=w["poly_params"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
               polygon=w["polygon"])
o.set_traj(w["coeffs"], w["T"])
o.penalty(w["points"], nthreads=os.cpu_count())
c = o.counters()
h = np.array(list(c["gd_pass_hist"]))
print(cfg, "points", P, "solves", c["solves"], "passes/solve %.2f" % (c["gd_passes"] / max(c["solves"], 1)),
      "trials/solve %.1f" % (c["gd_trials"] / max(c["solves"], 1)), "max passes", c["gd_max_passes"])
for i, n in enumerate(h):
    if n:
        print("  passes %3d..%3d%s  %9d  %.5f" % (4 * i, 4 * i + 3, "+" if i == 31 else " ", n, n / h.sum()))
