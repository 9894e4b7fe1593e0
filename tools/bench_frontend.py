"""Throughput of the batched front-end collision check on one MI355X vs the CPU oracle.

A "unit" is one (edge, obstacle point) pair = up to 50 interpolated SDF evaluations (sw_manager.hpp:1189).
The C-ABI call takes HOST buffers (the A* front end produces them per expansion), so the number includes
the PCIe upload of 16 B per pair and the flag read-back."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
import svsdf_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=20000)
ap.add_argument("--points", type=int, default=150, help="obstacle points per edge")
ap.add_argument("--shape", default="star")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--cpu-edges", type=int, default=400)
a = ap.parse_args()
rng = np.random.default_rng(20240807)
E = a.edges
fs = np.column_stack([rng.uniform(5, 25, E), rng.uniform(5, 70, E), rng.uniform(-np.pi, np.pi, E)])
cs = fs + np.column_stack([rng.integers(-1, 2, E), rng.integers(-1, 2, E), rng.uniform(-0.4, 0.4, E)])
# obstacle cells in the (kernel_size/2+1 = 9.5 m) box, most of them outside the robot like on the demo maps
pts = [fs[e, :2] + rng.uniform(-9.5, 9.5, (a.points, 2)) for e in range(E)]
keep = [q[np.hypot(*(q - fs[e, :2]).T) > 2.2] for e, q in enumerate(pts)]
ctx = svsdf_amd.SvsdfContext(shape=a.shape, device=0)
got = ctx.check_sub_sw_collision(fs, cs, keep)  # warm-up
pairs = sum(len(q) for q in keep)
t0 = time.perf_counter()
for _ in range(a.reps):
    got = ctx.check_sub_sw_collision(fs, cs, keep)
dt = (time.perf_counter() - t0) / a.reps
out = {"metric": "front-end collision check, (edge, obstacle point) pairs/s", "value": pairs / dt, "edges": E,
       "pairs": pairs, "ms_per_batch": 1e3 * dt, "free_fraction": float(got.mean()), "shape": a.shape,
       "note": "host buffers in, flags out (PCIe inclusive)"}
try:
    from oracle import orc
    o = orc.Oracle(a.shape)
    n = min(a.cpu_edges, E)
    t0 = time.perf_counter()
    want = np.array([o.check_sub_sw_collision(fs[e], cs[e], keep[e]) for e in range(n)])
    dtc = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": sum(len(q) for q in keep[:n]) / dtc, "unit": "pairs/s", "cores": 1,
                           "kind": "port", "sample": "first %d edges, oracle (early exit like the reference)" % n}
    out["parity"] = bool((want == got[:n]).all())
except Exception as ex:  # oracle is test infrastructure; the bench still reports the GPU number
    out["cpu_baseline"] = {"error": str(ex)}
print(json.dumps(out))
