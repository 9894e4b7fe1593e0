"""A/B of experimental builds against the default library on the 1 M-point workloads.

usage: python tools/exp_variants.py <variant>[,<variant>...] [config[,config...]] [points]
Each variant is implicit-svsdf-planner_amd/libsvsdf_hip_<variant>.so (tools build them with extra -D flags).
Runs every (library, config) in its own process (the library is chosen at import), prints ms per evaluation,
kernel split, and whether (cost, gradT, gradC) and the per-point results are bit-identical to the default build."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, time, hashlib
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "implicit-svsdf-planner_amd"))
import svsdf_amd
from svsdf_amd import workload
cfg, P, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                           poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                           tail_state=w["tail_state"], device=0)
c.set_points(w["points"])
for _ in range(16):   # until the launch plan (bound mode, widths, batch count) has settled
    out = c.eval_penalty(w["coeffs"], w["T"])
    if c.stats().get("plan_settled", 1):
        break
out = c.eval_penalty(w["coeffs"], w["T"])
t0 = time.perf_counter()
for _ in range(steps):
    out = c.eval_penalty(w["coeffs"], w["T"])
ms = 1e3 * (time.perf_counter() - t0) / steps
st = c.stats()
c.set_profiling(True)
c.eval_penalty(w["coeffs"], w["T"])
sp = c.stats()
c.set_profiling(False)
q = c.query_points(w["coeffs"], w["T"])
h = hashlib.sha256()
for a in (np.array([out[0]]), out[1], out[2], q[0], q[1], q[2]):
    h.update(np.ascontiguousarray(a).tobytes())
print(json.dumps(dict(ms=ms, solve_ms=sp["solve_ms"], device_ms=sp["device_ms"], solves=st["solves"], evals=st["sdf_evals"],
                      scan=st["scan_evals"], mode=st["gsip_bound_mode"], batches=st.get("batches"), spec=st.get("speculative_evals"), hash=h.hexdigest()[:16])))
'''


def run(variant, cfg, P, steps=10):
    env = dict(os.environ)
    if variant:
        env["SVSDF_LIB_VARIANT"] = variant
    else:
        env.pop("SVSDF_LIB_VARIANT", None)
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), cfg, str(P), str(steps)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if out.returncode:
        return dict(error=out.stderr.decode()[-400:])
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def main():
    variants = [v for v in sys.argv[1].split(",") if v] if len(sys.argv) > 1 else []
    cfgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C3", "NS"]
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
    for cfg in cfgs:
        run("", cfg, P, steps=3)      # (the first process on a fresh box pays for the driver's lazy initialisation: discarded)
        base = run("", cfg, P)
        print(f"{cfg:4s} {'default':12s} {json.dumps(base)}", flush=True)
        for v in variants:
            r = run(v, cfg, P)
            same = r.get("hash") == base.get("hash")
            print(f"{cfg:4s} {v:12s} {json.dumps(r)} identical={same} speedup={base.get('ms', 0) / r['ms'] if 'ms' in r else 0:.3f}", flush=True)


if __name__ == "__main__":
    main()
