"""Polygon shape on outlines of different sizes, one library variant against another.

usage: python tools/poly_outline_ab.py <variant>[,<variant>...] [mesh[,mesh...]] [points] [steps]
Every (library, mesh) runs in its own process ('-' = the default build); prints ms per evaluation and a hash over
(cost, gradT, gradC, per-point results)."""
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
mesh, P, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = workload.make(dict(shape="Polygon", mesh=mesh, scenario="star", N=16, P=P), minco=svsdf_amd.minco_coeffs)
t0 = time.perf_counter()
c = svsdf_amd.SvsdfContext(shape="Polygon", safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                           poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                           tail_state=w["tail_state"], device=0)
create_ms = 1e3 * (time.perf_counter() - t0)
c.set_points(w["points"])
for _ in range(8):
    out = c.eval_penalty(w["coeffs"], w["T"])
t0 = time.perf_counter()
for _ in range(steps):
    out = c.eval_penalty(w["coeffs"], w["T"])
ms = 1e3 * (time.perf_counter() - t0) / steps
q = c.query_points(w["coeffs"], w["T"])
h = hashlib.sha256()
for a in (np.array([out[0]]), out[1], out[2], q[0], q[1], q[2]):
    h.update(np.ascontiguousarray(a).tobytes())
print(json.dumps(dict(mesh=mesh, verts=len(w["polygon"]), ms=round(ms, 3), create_ms=round(create_ms, 1), hash=h.hexdigest()[:16])))
'''

if __name__ == "__main__":
    variants = ["-"] + [v for v in sys.argv[1].split(",") if v]
    meshes = (sys.argv[2] if len(sys.argv) > 2 else "star,sdHeart,sdRoundedCross,sdArc").split(",")
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 300000
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    for m in meshes:
        base = None
        for v in variants:
            env = dict(os.environ)
            env.pop("SVSDF_LIB_VARIANT", None)
            if v != "-":
                env["SVSDF_LIB_VARIANT"] = v
            try:
                r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), m, str(P), str(steps)], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
                d = json.loads(r.stdout.decode().strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001
                print(m, v, "failed:", e, r.stderr.decode()[-400:] if "r" in dir() else "")
                continue
            base = base or d
            print(f"{m:16s} {v:8s} {json.dumps(d)} identical={d['hash'] == base['hash']} speedup={base['ms'] / d['ms']:.3f}")
