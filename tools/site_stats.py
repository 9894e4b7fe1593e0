"""Lane occupancy of k_solve's four evaluation sites (table scan, layers 2-4, FD derivative, ladder).

usage: python tools/site_stats.py <variant> [config[,config...]] [points]
<variant> must be a -DSVSDF_SITE_STATS build (python implicit-svsdf-planner_amd/build.py --variant <v> -DSVSDF_SITE_STATS ...):
its k_solve counts, per site, the wave-level executions and the lanes that evaluate there; svsdf_debug_site_stats reads
the counters of the last evaluation."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SVSDF_LIB_VARIANT"] = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "implicit-svsdf-planner_amd"))
import svsdf_amd  # noqa: E402
from svsdf_amd import workload  # noqa: E402

cfgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C3", "NS"]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
for cfg in cfgs:
    if cfg.startswith("ref:"):   # reference-scale case (demo map through the producer, 24 pieces, generic durations)
        w = workload.reference_case(cfg[4:], N=24)
        T = svsdf_amd.forward_T(w["xs"][0][:24])
        w["T"] = T
        w["coeffs"] = svsdf_amd.minco_coeffs(w["head_state"], w["tail_state"], w["xs"][0][24:].reshape(-1, 3), T)
    else:
        w = workload.make(cfg, P=P, minco=svsdf_amd.minco_coeffs)
    c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                               poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                               tail_state=w["tail_state"], device=0)
    c.set_points(w["points"])
    for _ in range(16):
        c.eval_penalty(w["coeffs"], w["T"])
        if c.stats().get("plan_settled", 1):
            break
    c.eval_penalty(w["coeffs"], w["T"])
    out = (C.c_ulonglong * 26)()
    c.L.svsdf_debug_site_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    rc = c.L.svsdf_debug_site_stats(c.ctx, out)
    assert rc == 0, rc
    tot = sum(out[:4])
    print(f"== {cfg} lib={sys.argv[1]}  site executions {tot}")
    for i, name in enumerate(["table scan", "layers 2-4", "derivative", "ladder"]):
        ex, ln = out[i], out[4 + i]
        print(f"  {name:11s} executions {ex:12d} ({100.0 * ex / max(tot, 1):5.1f} %)  evaluating lanes {ln:13d}  "
              f"occupancy {ln / max(64 * ex, 1):.3f}")
    cyc = [out[8], out[9], out[10]]
    print("  wave cycles: scan %.3e  layers %.3e  descent %.3e  (shares %s); ladder steps with all groups open: %d" % (
        cyc[0], cyc[1], cyc[2], " / ".join("%.2f" % (c / max(sum(cyc), 1)) for c in cyc), out[11]))
    names = ["close", "candidate list", "samples + cheap bound", "seed scans", "selection", "flush", "whole wave", "staging"]
    tot = max(out[18], 1)
    print("  k_round wave cycles: " + ", ".join("%s %.3e (%.2f)" % (n, out[12 + i], out[12 + i] / tot) for i, n in enumerate(names)))
    print("  k_round seed-scan evaluation site: executions %d, evaluating lanes %d, occupancy %.3f" % (out[20], out[21], out[21] / max(64 * out[20], 1)))
    if out[22]:
        c.set_profiling(True); c.eval_penalty(w["coeffs"], w["T"]); sp = c.stats(); c.set_profiling(False)
        print("  k_tail's slowest wave: %d dependent evaluation steps (seed scans + solve passes) in %d cycles = %.0f cycles per step; "
              "k_tail %.1f us of a %.1f us device span (%.0f MHz)" % (out[22], out[23], out[23] / max(out[22], 1), 1e3 * sp["tail_ms"], 1e3 * sp["device_ms"], sp["shader_clock_mhz"]))
    st = c.stats()
    print("  ", {k: st[k] for k in ("solves", "sdf_evals", "scan_evals", "round_scan_evals", "gsip_bound_mode", "batches")})
