"""Compare library variants (strict vs fast) on GPU: timing and deviation."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
    import numpy as np, svsdf_amd
    from svsdf_amd import workload
    cfg, P = sys.argv[2], int(sys.argv[3])
    w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                 poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"], device=0)
    ctx.set_points(w["points"])
    sdf, ts, g, _ = ctx.query_points(w["coeffs"], w["T"])
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); out = ctx.eval_penalty(w["coeffs"], w["T"]); best = min(best, time.perf_counter() - t0)
    np.savez(sys.argv[4], sdf=sdf, ts=ts, g=g, cost=out[0], gT=out[1], gC=out[2], ms=best * 1e3, solve_ms=ctx.stats()["solve_ms"])
else:
    import numpy as np
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"; P = sys.argv[2] if len(sys.argv) > 2 else "100000"
    variants = sys.argv[3:] or ["", "fma"]
    res = {}
    for v in variants:
        env = dict(os.environ); env["SVSDF_LIB_VARIANT"] = v
        out = f"/tmp/ab_{v or 'default'}.npz"
        subprocess.check_call([sys.executable, __file__, "child", cfg, P, out], env=env, stderr=subprocess.DEVNULL)
        res[v] = np.load(out)
    ref = res[variants[0]]
    for v in variants:
        r = res[v]
        flips = int((np.abs(r["ts"] - ref["ts"]) > 1e-6).sum())
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        fl = np.abs(r["ts"] - ref["ts"]) > 1e-6
        ext = ref["sdf"] > 0
        dg = np.abs(r["g"] - ref["g"]).max(axis=1)
        print(f"   flips ext={int((fl&ext).sum())} int={int((fl&~ext).sum())}  max|dg| ext={dg[ext].max():.2e} int={dg[~ext].max():.2e} "
              f"n(|dg|>1e-4) ext={int((dg[ext]>1e-4).sum())} int={int((dg[~ext]>1e-4).sum())} max dsdf ext={np.abs(r['sdf']-ref['sdf'])[ext].max():.2e}")
        if fl.any():
            i = np.where(fl)[0][:5]
            for k in i: print("     pt", k, "sdf", ref["sdf"][k], r["sdf"][k], "t", ref["ts"][k], r["ts"][k], "g", ref["g"][k], r["g"][k])
        print(f"{v or 'default':8s} {float(r['ms']):8.3f} ms solve={float(r['solve_ms']):.3f}  flips={flips} dsdf={np.abs(r['sdf']-ref['sdf']).max():.2e} "
              f"dcost={abs(r['cost']-ref['cost'])/abs(ref['cost']):.2e} dgC={rel(r['gC'],ref['gC']):.2e} dgT={rel(r['gT'],ref['gT']):.2e}")
