"""Polygon crossing fast path (GPU): per-point (sdf, t*, grad) identical between library variants + timing."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
    import numpy as np, svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C5", P=int(sys.argv[2]), minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); out = ctx.eval_penalty(w["coeffs"], w["T"]); best = min(best, time.perf_counter() - t0)
    np.savez(sys.argv[3], sdf=sdf, ts=ts, g=g, cost=out[0], ms=best * 1e3)
else:
    import numpy as np
    P = sys.argv[1] if len(sys.argv) > 1 else "200000"
    res = {}
    for v in ("", "exp"):
        env = dict(os.environ); env["SVSDF_LIB_VARIANT"] = v
        out = f"/tmp/poly_{v or 'default'}.npz"
        subprocess.check_call([sys.executable, __file__, "child", P, out], env=env)
        res[v] = np.load(out)
    a, b = res[""], res["exp"]
    print("ms default", float(a["ms"]), "exp", float(b["ms"]))
    print("identical sdf", np.array_equal(a["sdf"], b["sdf"]), "t*", np.array_equal(a["ts"], b["ts"]), "grad", np.array_equal(a["g"], b["g"]),
          "cost rel", abs(float(a["cost"]) - float(b["cost"])) / abs(float(a["cost"])))
