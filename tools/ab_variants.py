"""A/B of two library builds (default vs SVSDF_LIB_VARIANT=exp) on GPU: per-point bit identity + timing."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
    import numpy as np, svsdf_amd
    from svsdf_amd import workload
    w = workload.make(sys.argv[2], P=int(sys.argv[3]), minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    tt = []
    for _ in range(8):
        t0 = time.perf_counter(); out = ctx.eval_penalty(w["coeffs"], w["T"]); tt.append(time.perf_counter() - t0)
    st = ctx.stats()
    np.savez(sys.argv[4], sdf=sdf, ts=ts, g=g, cost=out[0], ms=min(tt) * 1e3, med=float(np.median(tt)) * 1e3, evals=st["sdf_evals"], solves=st["solves"])
else:
    import numpy as np
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"; P = sys.argv[2] if len(sys.argv) > 2 else "100000"
    res = {}
    for v in ("", "exp"):
        env = dict(os.environ); env["SVSDF_LIB_VARIANT"] = v
        out = f"/tmp/ab_{v or 'default'}.npz"
        subprocess.check_call([sys.executable, __file__, "child", cfg, P, out], env=env)
        res[v] = np.load(out)
    a, b = res[""], res["exp"]
    print(cfg, P, "ms default %.3f (med %.3f) exp %.3f (med %.3f)" % (float(a["ms"]), float(a["med"]), float(b["ms"]), float(b["med"])),
          "evals", int(a["evals"]), int(b["evals"]))
    print("   identical sdf", np.array_equal(a["sdf"], b["sdf"]), "t*", np.array_equal(a["ts"], b["ts"]), "grad", np.array_equal(a["g"], b["g"]),
          "cost rel", abs(float(a["cost"]) - float(b["cost"])) / abs(float(a["cost"])))
