"""Why is a workload slower as a sub-run of bench.py than in its own process?  (VERDICT r4 'weak #4': NS 7.09 ms alone,
7.6 - 7.7 ms after the C3 headline in the same process.)

usage: python tools/state_probe.py [sequence ...]      (default: all)
Every sequence runs in its OWN process; the NS block is the same in all of them: context, settle, 3 x 20 timed
evaluations, one profiled evaluation (HIP-event device span, merged k_solve / k_round time), the shader clock the kernels
measured themselves, the plan, free device memory.
  alone        NS in a fresh process (GPU_MAX_HW_QUEUES=8 like bench.py)
  alone_q4     the same with the runtime's default number of hardware queues
  after_c3     C3 context: settle + 60 evaluations, closed; then NS
  beside_c3    C3 context kept open (idle) while NS runs
  twice        NS, closed, NS again
  churn        20 GB of device memory allocated and freed through torch first, then NS
  after_idle   C3 as in after_c3, then 3 s of sleep, then NS   (clock / thermal recovery)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, time
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "implicit-svsdf-planner_amd"))
seq = sys.argv[1]
import torch
import svsdf_amd
from svsdf_amd import workload

def make(cfg):
    w = workload.make(cfg, minco=svsdf_amd.minco_coeffs)
    c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                               poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                               tail_state=w["tail_state"], device=0)
    c.set_points(w["points"])
    for _ in range(16):
        c.eval_penalty(w["coeffs"], w["T"])
        if c.stats().get("plan_settled", 1):
            break
    c.eval_penalty(w["coeffs"], w["T"])
    return w, c

def block(tag):
    free0 = torch.cuda.mem_get_info()[0]
    w, c = make("NS")
    reps = []
    clk = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            c.eval_penalty(w["coeffs"], w["T"])
        torch.cuda.synchronize()
        reps.append(1e3 * (time.perf_counter() - t0) / 20)
        clk.append(c.stats().get("shader_clock_mhz", 0.0))
    c.set_profiling(True)
    c.eval_penalty(w["coeffs"], w["T"])
    c.eval_penalty(w["coeffs"], w["T"])
    sp = c.stats()
    c.set_profiling(2)
    c.eval_penalty(w["coeffs"], w["T"])
    c.eval_penalty(w["coeffs"], w["T"])
    ss = c.stats()
    c.set_profiling(False)
    pl = c.get_plan()
    out = dict(seq=seq, tag=tag, ms=[round(x, 3) for x in reps], clock_mhz=[round(x) for x in clk], device_ms=round(sp["device_ms"], 3),
               solve_ms=round(sp["solve_ms"], 3), round_ms=round(sp["round_ms"], 3), serial_device_ms=round(ss["device_ms"], 3),
               serial_solve_ms=round(ss["solve_ms"], 3), serial_round_ms=round(ss["round_ms"], 3),
               plan=pl, interior=sp["interior_points"], solves=sp["solves"], evals=sp["sdf_evals"], round_scan=sp["round_scan_evals"],
               free_gb_before=round(free0 / 2**30, 2), free_gb_now=round(torch.cuda.mem_get_info()[0] / 2**30, 2),
               hwq=os.environ.get("GPU_MAX_HW_QUEUES"))
    print(json.dumps(out), flush=True)
    return w, c

def c3(n=60):
    w, c = make("C3")
    t0 = time.perf_counter()
    for _ in range(n):
        c.eval_penalty(w["coeffs"], w["T"])
    print(json.dumps(dict(seq=seq, tag="C3", ms=round(1e3 * (time.perf_counter() - t0) / n, 3), clock_mhz=round(c.stats().get("shader_clock_mhz", 0.0)))), flush=True)
    return w, c

if seq in ("alone", "alone_q4"):
    block("NS")
elif seq == "after_c3":
    w, c = c3(); c.close(); block("NS after C3")
elif seq == "beside_c3":
    w, c = c3(); block("NS beside C3")
elif seq == "twice":
    w, c = block("NS #1"); c.close(); block("NS #2")
elif seq == "churn":
    xs = [torch.empty(int(2.5 * 2**30), dtype=torch.uint8, device="cuda") for _ in range(8)]
    for x in xs: x.fill_(1)
    torch.cuda.synchronize(); del xs; torch.cuda.empty_cache()
    block("NS after 20 GB churn")
elif seq == "after_idle":
    w, c = c3(); c.close(); time.sleep(3.0); block("NS after C3 + 3 s idle")
'''

def main():
    seqs = sys.argv[1:] or ["alone", "alone_q4", "after_c3", "beside_c3", "twice", "churn", "after_idle"]
    for s in seqs:
        env = dict(os.environ)
        if s == "alone_q4":
            env.pop("GPU_MAX_HW_QUEUES", None)
        else:
            env["GPU_MAX_HW_QUEUES"] = "8"
        try:
            out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), s], env=env, stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE, timeout=300)
            sys.stdout.write(out.stdout.decode())
            if out.returncode:
                print(json.dumps(dict(seq=s, error=out.stderr.decode()[-500:])))
        except subprocess.TimeoutExpired:
            print(json.dumps(dict(seq=s, error="timeout")))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
