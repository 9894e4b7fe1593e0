"""A/B harness (GPU): solver variants must give bit-identical per-point results; prints timings."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
import numpy as np
import svsdf_amd
from svsdf_amd import workload

def run(w, env, reps=3):
    for k, v in env.items():
        os.environ[k] = str(v)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"],
                                 head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = ctx.eval_penalty(w["coeffs"], w["T"]); best = min(best, time.perf_counter() - t0)
    st = ctx.stats()
    ctx.close()
    return (sdf, ts, g, out), best, st

if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else None
    w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
    base = dict(SVSDF_G=1, SVSDF_G_LATE=1, SVSDF_PRUNE=0, SVSDF_BATCHES=1, SVSDF_BLOCK=64)
    ref, t, st = run(w, base)
    print(f"base {base}: {t*1e3:8.3f} ms  refine_ms={st['solve_ms']:.3f} dev_ms={st['device_ms']:.3f} evals={st['sdf_evals']} scan={st['scan_evals']} interior={st['interior_points']}")
    variants = []
    for g, gl in ((1, 1), (2, 2), (4, 4), (4, 8), (8, 8), (2, 8)):
        for nb in (1, 4):
            variants.append(dict(SVSDF_G=g, SVSDF_G_LATE=gl, SVSDF_PRUNE=1, SVSDF_BATCHES=nb, SVSDF_BLOCK=64))
    variants.append(dict(SVSDF_G=4, SVSDF_G_LATE=8, SVSDF_PRUNE=1, SVSDF_BATCHES=8, SVSDF_BLOCK=64))
    variants.append(dict(SVSDF_G=4, SVSDF_G_LATE=8, SVSDF_PRUNE=1, SVSDF_BATCHES=4, SVSDF_BLOCK=256))
    for v in variants:
        r, t, st = run(w, v)
        same = all(np.array_equal(a, b) for a, b in zip(r[:3], ref[:3]))
        dc = abs(r[3][0] - ref[3][0]) / abs(ref[3][0])
        print(f"G={v['SVSDF_G']} late={v['SVSDF_G_LATE']} nb={v['SVSDF_BATCHES']} blk={v['SVSDF_BLOCK']}: {t*1e3:8.3f} ms refine_ms={st['solve_ms']:.3f} dev_ms={st['device_ms']:.3f} evals={st['sdf_evals']} scan={st['scan_evals']} identical={same} dcost={dc:.2e}")
