#!/usr/bin/env python3
"""bench.py -- SVSDF query-points/sec (cost+grad) per optimizer evaluation on MI355X.

One "step" = one call of the inner operator addSaftyPenaOnSweptVolumeParallelTrueSDF (reference
BEO:774-869) over the whole query-point cloud, points already resident in HBM, through the C ABI
(include/svsdf_c.h).

Workloads (svsdf_amd/workload.py, BASELINE.json `configs`):
  --gpus 1 (default)  C3 = configs[2]: sdHorseshoe, 32-piece MINCO, 1 M corridor query points -- the largest
                      single-GPU configuration (BASELINE.json quotes its metric on no particular config).
                      The same line carries, as sub-objects measured in the same run: "north_star" (star,
                      16 pieces, 1 M points: the workload BASELINE.json's north_star target is quoted on),
                      "map_distribution" (the headline config on the map-uniform cloud) and the
                      full-callback time (a14: MINCO forward + penalty + adjoint).
  --gpus N > 1        C4 = configs[3]: sdHeart, 32 pieces, 4 M points in total, striped over the N GPUs
                      (strong scaling), one sum of the (19N+1)-double partial per evaluation:
                        launched by torchrun (WORLD_SIZE set), one process per GPU -> RCCL all-reduce
                                                           (torch.distributed), or
                        plain `python bench.py --gpus N` -> ONE process drives devices 0..N-1 through the C ABI's
                                                           multi-device context (host combine or in-process RCCL,
                                                           --combine); refuses to run when fewer than N GPUs are
                                                           visible unless --devices repeats ordinals on purpose.
  --config / --points / --dist override the workload (C1..C5, NS).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (DESIGN.md "Measurement" explains every field).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# The library can run a large shard as up to 4 point batches on their own HIP streams; streams that share one of the
# runtime's hardware queues (default 4) serialise.  This is the HOST's setting: it must be in the
# environment before the HIP runtime initialises, i.e. before `import torch`.  The library itself never touches it -- it
# times 1 / 3 / 4 batches on the first evaluations after svsdf_set_points and keeps the fastest (svsdf_stats.batches).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
FP64_PEAK_TFLOPS = 78.6    # FP64 vector peak = 1/2 of the 157.3 TF FP32 vector figure
BYTES_PER_POINT = 24.0     # algorithmic bytes per query point per evaluation (3 x f64)
FP64_NOFMA_TFLOPS = 39.3   # the same VALU rate without FMA (the parity build is -ffp-contract=off: one flop per lane-op)
FP64_NOFMA_MEASURED_TFLOPS = 31.6   # what this part sustains on v_mul_f64 + v_add_f64 chains (tools/experiments/fp64_peak.hip,
                                    # profiles/r03_fp64_peak.txt); reported beside the nominal figures, never instead of them
FLOP_PER_EVAL = 150.0      # nominal FP64 flop per full SDF-at-time evaluation: polynomial 36 + sincos ~60 + transform 10 +
                           # shape ~45
FLOP_PER_TABLE_EVAL = 55.0 # a layer-1 TABLE evaluation (pose from the LDS table): transform 10 + shape ~45 only


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, help="workload (C1..C5, NS); default C3 at 1 GPU, C4 at N > 1")
    ap.add_argument("--points", type=int, default=None, help="TOTAL query points (default: the config's)")
    ap.add_argument("--dist", default="corridor", choices=["corridor", "map"])
    ap.add_argument("--inprocess", action="store_true",
                    help="one process drives --gpus devices through the C ABI's multi-device context (the default "
                         "for --gpus N > 1 when not launched by torchrun)")
    ap.add_argument("--devices", default=None, help="comma list of HIP ordinals for --inprocess (default 0..N-1; "
                    "an ordinal may repeat to emulate several stripes on one GPU)")
    ap.add_argument("--combine", default="auto", choices=["auto", "host", "rccl"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (rank 0, 1 GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the north_star / map_distribution sub-runs")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--sustained-steps", type=int, default=300, help="length of the `sustained` run (generic durations)")
    ap.add_argument("--only", default=None, help="run ONE sub-measurement and print its object: reference_scale")
    return ap.parse_args()


def cpu_baseline(w, budget_s):
    """Oracle (CPU restatement of the reference path, OpenMP schedule(dynamic) over points like BEO:785) timed on
    the host cores on a bounded prefix sample of the same workload: one probe, then 3 timed runs of a sample
    sized to a third of the budget each; the median is reported and scales linearly in the number of points
    (independent points, uniformly random order)."""
    from oracle import orc
    cores = os.cpu_count() or 1
    o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                   poly_params=w["poly_params"], polygon=w["polygon"],
                   head_state=w["head_state"], tail_state=w["tail_state"])
    o.set_traj(w["coeffs"], w["T"])
    pts = w["points"]
    probe = min(len(pts), max(512, 16 * cores))
    t0 = time.perf_counter()
    o.penalty(pts[:probe], nthreads=cores)
    tp = time.perf_counter() - t0
    n = int(min(len(pts), max(probe, probe * (budget_s / 3.0) / max(tp, 1e-6))))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        o.penalty(pts[:n], nthreads=cores)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    cnt = o.counters()
    return {"value": n / t, "unit": "query-points/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of {len(pts)} query points of the same workload (random order, so a prefix is a "
                      f"uniform subsample; rate scales linearly in points), oracle/liborc.so (-O3 -fopenmp, "
                      f"schedule(dynamic), {cores} threads), median of 3 runs {sorted(round(x, 3) for x in ts)} s "
                      f"after 1 probe run",
            "sdf_evals_per_point": cnt["sdf_evals"] / max(n, 1)}


class Runner:
    """One workload resident on this rank's device(s)."""

    def __init__(self, a, name, P_total, dist_name, rank, world, local_rank, tdist, devices):
        import svsdf_amd
        from svsdf_amd import workload
        self.a, self.name, self.tdist, self.world = a, name, tdist, world
        self.w = w = workload.make(name, P=P_total, dist=dist_name, minco=svsdf_amd.minco_coeffs)
        self.N = N = len(w["T"])
        self.P_total = P_total
        self.opt = opt = svsdf_amd.TrajOptimizer()
        combine = {"auto": 0, "host": 1, "rccl": 2}[a.combine]
        opt.setParam(dict(rho=w["rho"], weight_p=w["weight_p"], safety_hor=w["safety_hor"],
                          inputdata=f"shapes/{w['shape']}.obj", poly_params=w["poly_params"],
                          polygon=w["polygon"], device=local_rank, devices=devices, combine=combine))
        opt.setConditions(w["head_state"], w["tail_state"], N)
        t0 = time.perf_counter()
        opt.setPoints(w["points"])          # uploaded once; each device keeps its stripe in HBM
        self.ctx = opt._context()
        self.set_points_ms = 1e3 * (time.perf_counter() - t0)
        self.x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
        self.zT, self.zC = np.zeros(N), np.zeros((6 * N, 3))
        self.n_eval = 0      # device evaluations issued by this runner (tools/profile_round.sh divides counters by it)

    def step(self):
        self.n_eval += 1
        return self.opt.addSaftyPenaOnSweptVolumeParallelTrueSDF(self.w["T"], self.w["coeffs"], 0.0, self.zT, self.zC)

    def fence(self):
        import torch
        if self.tdist is not None:
            self.tdist.barrier()
        torch.cuda.synchronize()

    def settle(self):
        """Setup after set_points, outside every timed region: the first evaluations decide the GSIP bound mode
        (deterministic rule, svsdf_stats.bound_ratio), learn the launch widths and -- for a large shard in a scanning
        mode -- time 1 / 3 / 4 point batches once each (svsdf_stats.plan_settled; all of them return identical results)."""
        t0 = time.perf_counter()
        self.step()
        first_ms = 1e3 * (time.perf_counter() - t0)
        n = 1
        while n < 8:
            self.step()
            n += 1
            if self.ctx.stats()["plan_settled"]:
                break
        st = self.ctx.stats()
        return {"set_points_ms": self.set_points_ms, "set_points_library_ms": st["setup_ms"],
                "first_evaluation_ms": first_ms, "settle_evaluations": n, "batches": st["batches"],
                "gsip_bound_mode": ["cheap-chunk", "full-scan", "lazy-scan", "anchor-scan"][st["gsip_bound_mode"]],
                "bound_ratio": st["bound_ratio"], "rule": "deterministic, from the first evaluation's counters: GSIP solves / "
                "samples > 0.5 (Polygon 0.2) -> full-scan; else lazy-scan from 400 k points per device, cheap-chunk below (a cloud small enough for the fused tail tries lazy-scan once and keeps it when it saves a quarter of the GSIP solves); "
                "batches: 1 / 3 / 4 timed once each on a large shard in a scanning mode, fastest kept"}

    def timed(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        self.fence()
        acc = dict(sdf_evals=0, scan_evals=0, solves=0, solve_launches=0, gsip_samples=0, round_scan_evals=0, speculative_evals=0)
        last = None
        combine_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
            last = self.ctx.stats()
            for k in acc:
                acc[k] += last[k]
            combine_ms += last["combine_ms"]
        self.fence()
        elapsed = time.perf_counter() - t0
        return elapsed, acc, last, combine_ms / steps

    def profiled(self, steps):
        """Kernel times of k_solve and k_round: separate passes with per-launch HIP events on the library's own
        streams (an event record costs ~6 us per launch, so it stays out of the timed region)."""
        self.ctx.set_profiling(True)
        solve_ms = dev_ms = self.solve_ms_sum = self.round_ms = self.round_ms_sum = 0.0
        for _ in range(steps):
            self.step()
            st = self.ctx.stats()
            solve_ms += st["solve_ms"]
            dev_ms += st["device_ms"]
            self.solve_ms_sum += st["solve_ms_sum"] / steps
            self.round_ms += st["round_ms"] / steps
            self.round_ms_sum += st["round_ms_sum"] / steps
        # the same with the point batches run one after the other: a launch's duration is then its own cost
        self.ctx.set_profiling(2)
        self.solve_ms_serial = self.dev_ms_serial = self.round_ms_serial = 0.0
        self.step()
        for _ in range(steps):
            self.step()
            st = self.ctx.stats()
            self.solve_ms_serial += st["solve_ms"] / steps
            self.round_ms_serial += st["round_ms"] / steps
            self.dev_ms_serial += st["device_ms"] / steps
        self.ctx.set_profiling(False)
        self.step()
        return solve_ms / steps, dev_ms / steps

    def generic_durations(self, steps, seed=11):
        """The regime an optimisation is in after its first callback: piece durations are generic doubles (here every
        tau gets a fixed relative perturbation of 1e-3 N(0,1), the waypoints stay), so the reference's chain of
        subtractions for the piece-local time runs (svsdf_stats.piece_time_exact = 1).  Timed exactly like the headline:
        `steps` evaluations of the inner operator on ONE fixed perturbed trajectory, fenced on both sides."""
        import svsdf_amd
        rng = np.random.default_rng(seed)
        N = self.N
        x = self.x.copy()
        x[:N] *= 1.0 + 1e-3 * rng.standard_normal(N)
        T = svsdf_amd.forward_T(x[:N])
        coeffs = svsdf_amd.minco_coeffs(self.w["head_state"], self.w["tail_state"], x[N:].reshape(-1, 3), T)
        f = lambda: self.opt.addSaftyPenaOnSweptVolumeParallelTrueSDF(T, coeffs, 0.0, self.zT, self.zC)
        self.n_eval += 3 + steps
        for _ in range(3):
            f()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            f()
        self.fence()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        st = self.ctx.stats()
        for _ in range(2):
            self.step()     # back on the headline trajectory (launch plan)
        return {"ms_per_step": ms, "value": self.P_total / (ms * 1e-3), "unit": "query-points/s", "steps": steps,
                "piece_time_exact": st["piece_time_exact"],
                "note": "same workload, piece durations 2.5 s * (1 + ~1e-3 N(0,1)) (generic doubles): the reference-faithful "
                        "chain of subtractions for the piece-local time runs; this is what every LMBM iterate after the "
                        "first one sees"}

    def full_callback(self, steps, perturb=0.0, seed=7):
        """a14: costFunctionLmbmParallel (tau -> T, MINCO forward, penalty, adjoint, chain rule).  With
        perturb > 0 every call sees a slightly different x (relative size `perturb`), like the successive
        callbacks of an optimisation: the launch plan learned from the previous call no longer matches exactly."""
        rng = np.random.default_rng(seed)
        xs = [self.x * (1.0 + perturb * rng.standard_normal(len(self.x))) if perturb else self.x for _ in range(steps)]
        self.n_eval += 1 + steps
        self.opt.costFunctionLmbmParallel(self.x)
        self.fence()
        t0 = time.perf_counter()
        for x in xs:
            self.opt.costFunctionLmbmParallel(x)
        self.fence()
        return 1e3 * (time.perf_counter() - t0) / steps


def combine_ab(r, a, devices, steps):
    """In-process multi-device context: `steps` evaluations under the host combine and under the RCCL combine, same
    context, same resident stripes; the communicator's rank count is read back from RCCL (ncclCommCount)."""
    out = {"steps": steps}
    distinct = len(set(devices)) == len(devices)
    for mode in ("host", "rccl"):
        if mode == "rccl" and not distinct:
            out["rccl_ms_per_step"] = None
            out["rccl_ranks"] = 0
            out["rccl_note"] = "--devices repeats an ordinal (several stripes on one GPU): an RCCL communicator needs one rank per distinct GPU; skipped"
            continue
        try:
            r.ctx.set_combine(mode)
        except Exception as e:      # librccl missing / init failure: say so, do not fake a number
            out[f"{mode}_ms_per_step"] = None
            out["rccl_note"] = f"RCCL combine unavailable: {e}"
            continue
        try:
            for _ in range(2):
                r.step()
            r.fence()
            cm = 0.0
            t0 = time.perf_counter()
            for _ in range(steps):
                r.step()
                cm += r.ctx.stats()["combine_ms"]
            r.fence()
            out[f"{mode}_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / steps
            out[f"{mode}_combine_only_ms"] = cm / steps
            if mode == "rccl":
                out["rccl_ranks"] = r.ctx.group_info()["rccl_ranks"]
        except Exception as e:      # an all-reduce that fails must not take the headline measurement with it
            out[f"{mode}_ms_per_step"] = None
            out["rccl_note"] = f"{mode} combine failed: {e}"
            break
    out["note"] = ("whole-evaluation time per step under each combine (same context, same stripes); *_combine_only_ms: host time "
                   "from 'all devices done' to 'summed partial on the host'; rccl_ranks: ncclCommCount of the communicator")
    try:
        r.ctx.set_combine({"auto": "host", "host": "host", "rccl": "rccl"}[a.combine])
    except Exception:
        pass
    return out


def allreduce_ms(tdist, n, reps=50):
    """The collective alone: RCCL all-reduce of n doubles (device buffer, in place), HIP-event timed."""
    import torch
    t = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(5):
        tdist.all_reduce(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tdist.all_reduce(t)
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def fp64_accounting(acc, steps, solve_ms, solve_ms_serial, round_ms, round_ms_serial, ms_per_step, ndev=1):
    """FP64 figures per kernel from the kernels' own counters (svsdf_stats): k_solve = its full evaluations x 150 flop +
    its table evaluations x 55 flop over ITS time; k_round = its table evaluations x 55 flop over ITS time; the whole
    evaluation = all of it over the driver-visible ms_per_step.  Fractions of the 78.6 TFLOP/s vector peak (FMA = 2 flop)
    and, beside it, of the 39.3 TFLOP/s the no-FMA parity build can reach at best."""
    e_tab = acc["scan_evals"] / steps / ndev
    e_full = (acc["sdf_evals"] - acc["scan_evals"]) / steps / ndev
    e_round = acc["round_scan_evals"] / steps / ndev
    fl_solve = e_full * FLOP_PER_EVAL + e_tab * FLOP_PER_TABLE_EVAL
    fl_round = e_round * FLOP_PER_TABLE_EVAL
    tf = lambda fl, ms: fl / max(ms, 1e-9) / 1e9          # flop / ms -> TFLOP/s
    def obj(fl, ms):
        return {"achieved": tf(fl, ms), "frac": tf(fl, ms) / FP64_PEAK_TFLOPS, "frac_of_no_fma_ceiling": tf(fl, ms) / FP64_NOFMA_TFLOPS,
                "ms": ms, "gflop": fl / 1e9}
    return {"bound": "fp64_valu", "peak": FP64_PEAK_TFLOPS, "peak_no_fma": FP64_NOFMA_TFLOPS,
            "peak_no_fma_measured": FP64_NOFMA_MEASURED_TFLOPS, "unit": "TFLOP/s",
            "flop_per_full_eval_nominal": FLOP_PER_EVAL, "flop_per_table_eval_nominal": FLOP_PER_TABLE_EVAL,
            "k_solve_full_evals_per_step": e_full, "k_solve_table_evals_per_step": e_tab, "k_round_table_evals_per_step": e_round,
            "k_solve_speculative_evals_per_step": acc["speculative_evals"] / steps / ndev,
            "speculative_note": "of the full evaluations: halving-ladder candidates evaluated behind the accepted one (G per "
                                "step; the reference's sequential loop would not evaluate them); counted as executed work",
            "k_solve": obj(fl_solve, solve_ms), "k_solve_serialized": obj(fl_solve, solve_ms_serial),
            "k_round": obj(fl_round, round_ms), "k_round_serialized": obj(fl_round, round_ms_serial),
            "whole_evaluation": obj(fl_solve + fl_round, ms_per_step),
            # headline fraction of this object: k_solve's own work over k_solve's own (merged) time
            "achieved": tf(fl_solve, solve_ms), "frac": tf(fl_solve, solve_ms) / FP64_PEAK_TFLOPS,
            "note": "per kernel: evaluations counted by that kernel / that kernel's HIP-event time (merged intervals of its "
                    "concurrent launches; *_serialized: batches run one after the other); table evaluations skip the "
                    "polynomial and sincos (55 of the nominal 150 flop); no FMA by policy, so 39.3 TFLOP/s is the ceiling"}


def sub_run(a, name, P, dist_name, rank, world, local_rank, tdist, devices, steps):
    r = Runner(a, name, P, dist_name, rank, world, local_rank, tdist, devices)
    setup = r.settle()
    elapsed, acc, last, _ = r.timed(steps, 2)
    solve_ms, dev_ms = r.profiled(min(steps, 3))
    ms_step = 1e3 * elapsed / steps
    fp = fp64_accounting(acc, steps, solve_ms, r.solve_ms_serial, r.round_ms, r.round_ms_serial, ms_step)
    out = {"workload": f"{name}: {r.w['shape']}, {r.N} pieces, {P} {dist_name} points", "points_total": P,
           "steps": steps, "ms_per_step": ms_step, "value": P * steps / elapsed,
           "unit": "query-points/s", "interior_fraction": last["interior_points"] / max(last["points"], 1),
           "culled_fraction": last["culled_points"] / max(last["points"], 1),
           "argmin_solves_per_point": acc["solves"] / steps / max(last["points"], 1),
           "gsip_bound_mode": setup["gsip_bound_mode"], "batches": setup["batches"], "k_solve_ms_per_step": solve_ms,
           "k_round_ms_per_step": r.round_ms,
           "fp64_frac_k_solve": fp["k_solve"]["frac"], "fp64_frac_k_round": fp["k_round"]["frac"],
           "fp64_frac_whole_evaluation": fp["whole_evaluation"]["frac"],
           "hbm_frac": BYTES_PER_POINT * last["points"] / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "generic_durations": r.generic_durations(steps),
           "full_callback_ms": r.full_callback(min(steps, 5))}
    r.opt._ctx.close()
    return out


def quick_run(name, P, steps, local_rank, generic=False):
    """Compact sub-measurement of another BASELINE config on one GPU: resident cloud, settle, `steps` timed evaluations
    of the inner operator (fenced), the same way the headline is timed."""
    import torch
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(name, P=P, minco=svsdf_amd.minco_coeffs)
    c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                               poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                               tail_state=w["tail_state"], device=local_rank)
    t0 = time.perf_counter()
    c.set_points(w["points"])
    set_ms = 1e3 * (time.perf_counter() - t0)
    n_settle = 0
    while n_settle < 8:
        c.eval_penalty(w["coeffs"], w["T"])
        n_settle += 1
        if c.stats()["plan_settled"] and n_settle >= 2:
            break
    for _ in range(2):
        c.eval_penalty(w["coeffs"], w["T"])
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        c.eval_penalty(w["coeffs"], w["T"])      # (every entry point synchronises before it returns)
        per.append(1e3 * (time.perf_counter() - t1))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    st, pl = c.stats(), c.get_plan()
    out = {"workload": f"{name}: {w['shape']}, {len(w['T'])} pieces, {P} corridor points", "points_total": P, "steps": steps,
           "ms_per_step": ms, "ms_per_step_median": float(np.median(per)), "value": P / (ms * 1e-3), "unit": "query-points/s",
           "set_points_ms": set_ms,
           "settle_evaluations": n_settle, "interior_fraction": st["interior_points"] / max(st["points"], 1),
           "plan": {"gsip_bound_mode": ["cheap-chunk", "full-scan", "lazy-scan", "anchor-scan"][pl["bound_mode"]], "batches": pl["batches"],
                    "lanes_per_query": pl["lanes_per_query"], "tail_iter": st["tail_iter"]}}
    if generic:
        rng = np.random.default_rng(11)
        N = len(w["T"])
        x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
        x[:N] *= 1.0 + 1e-3 * rng.standard_normal(N)
        T = svsdf_amd.forward_T(x[:N])
        coeffs = svsdf_amd.minco_coeffs(w["head_state"], w["tail_state"], x[N:].reshape(-1, 3), T)
        for _ in range(2):
            c.eval_penalty(coeffs, T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            c.eval_penalty(coeffs, T)
        torch.cuda.synchronize()
        out["generic_durations_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / steps
    c.close()
    return out


def first_optimisation(name, P, local_rank, callbacks=30, seed=5):
    """What optimize_traj_lmbm pays from a cold context (back_end_optimizer.cpp:29-36): svsdf_set_points, then
    `callbacks` full callbacks (a14) whose x differ from call to call like an optimiser's iterates (relative 1e-3), the
    plan-deciding evaluations included.  Wall time of the whole sequence and of its parts."""
    import torch
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(name, P=P, minco=svsdf_amd.minco_coeffs)
    N = len(w["T"])
    opt = svsdf_amd.TrajOptimizer()
    opt.setParam(dict(rho=w["rho"], weight_p=w["weight_p"], safety_hor=w["safety_hor"], inputdata=f"shapes/{w['shape']}.obj",
                      poly_params=w["poly_params"], polygon=w["polygon"], device=local_rank))
    opt.setConditions(w["head_state"], w["tail_state"], N)
    x0 = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
    rng = np.random.default_rng(seed)
    xs = [x0 * (1.0 + 1e-3 * rng.standard_normal(len(x0))) for _ in range(callbacks)]
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    opt.setPoints(w["points"])
    ctx = opt._context()          # creates the context and uploads the cloud
    t_set = time.perf_counter() - t_all
    per = []
    settled_at = None
    for k, x in enumerate(xs):
        t0 = time.perf_counter()
        opt.costFunctionLmbmParallel(x)
        per.append(1e3 * (time.perf_counter() - t0))
        if settled_at is None and ctx.stats()["plan_settled"]:
            settled_at = k + 1
    torch.cuda.synchronize()
    total = 1e3 * (time.perf_counter() - t_all)
    pl = ctx.get_plan()
    opt._ctx.close()
    return {"workload": f"{name}, {P} points, cold context", "callbacks": callbacks, "total_ms": total,
            "set_points_and_create_ms": 1e3 * t_set, "first_callback_ms": per[0], "second_callback_ms": per[1],
            "mean_callback_ms": float(np.mean(per)), "steady_callback_ms": float(np.mean(per[-10:])),
            "callbacks_until_plan_settled": settled_at,
            "plan": {"gsip_bound_mode": ["cheap-chunk", "full-scan", "lazy-scan", "anchor-scan"][pl["bound_mode"]], "batches": pl["batches"],
                     "lanes_per_query": pl["lanes_per_query"]},
            "note": "wall time of svsdf_create + svsdf_set_points + 30 costFunctionLmbmParallel calls with x * (1 + 1e-3 N(0,1)) "
                    "each (generic piece durations: the reference-faithful piece time runs from the first call on); the plan "
                    "follows deterministic rules (no timing inside the library)"}


def reference_scale(local_rank, calls=200, pieces=24, oracle_reps=5):
    """The regime the reference actually runs (VERDICT r4 'missing #2'): its three demo maps through the query-point
    producer (10^2 - 10^3 points), 24 MINCO pieces, generic durations, the FULL callback (a14, lmbm_evaluate_t) -- median
    of `calls` calls cycling through 8 slightly different iterates, timed around the foreign call itself (arguments
    marshalled once), with the oracle's callback beside it at the shipped yaml's threads_num (12) and on all host cores."""
    import ctypes as C
    import svsdf_amd
    from svsdf_amd import workload
    from oracle import orc
    cores = os.cpu_count() or 1
    out = {"pieces": pieces, "calls": calls, "cases": {}}
    for name in ("star", "sdHorseshoe", "sdHeart"):
        w = workload.reference_case(name, N=pieces)
        kw = dict(safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], poly_params=w["poly_params"],
                  head_state=w["head_state"], tail_state=w["tail_state"])
        c = svsdf_amd.SvsdfContext(shape=w["shape"], device=local_rank, **kw)
        c.set_points(w["points"])
        xs = [np.ascontiguousarray(x, dtype=np.float64) for x in w["xs"]]
        n = len(xs[0])
        g = np.zeros(n)
        dp = C.POINTER(C.c_double)
        xp = [x.ctypes.data_as(dp) for x in xs]
        gp = g.ctypes.data_as(dp)
        f = c.L.svsdf_lmbm_evaluate
        for k in range(2 * len(xs)):                       # plan settles, every iterate seen once
            f(c.ctx, xp[k % len(xs)], gp, n)
        per = np.zeros(calls)
        t_all = time.perf_counter()
        for k in range(calls):
            t0 = time.perf_counter_ns()
            f(c.ctx, xp[k % len(xs)], gp, n)
            per[k] = 1e-3 * (time.perf_counter_ns() - t0)
        wall = time.perf_counter() - t_all
        st, pl = c.stats(), c.get_plan()
        c.set_profiling(True)
        f(c.ctx, xp[0], gp, n)
        dev_us = 1e3 * c.stats()["device_ms"]
        c.set_profiling(False)
        fh, gh = c.lmbm_evaluate(xs[0])
        case = {"map_points": w["map_points"], "query_points": int(len(w["points"])),
                "interior_points": int(st["interior_points"]), "culled_points": int(st["culled_points"]),
                "callback_us_median": float(np.median(per)), "callback_us_p10": float(np.percentile(per, 10)),
                "callback_us_p90": float(np.percentile(per, 90)), "callback_us_mean": float(per.mean()),
                "callbacks_per_s": calls / wall, "device_us": dev_us,
                "piece_time_exact": int(st["piece_time_exact"]), "shader_clock_mhz": st.get("shader_clock_mhz", 0.0),
                "plan": {"gsip_bound_mode": ["cheap-chunk", "full-scan", "lazy-scan", "anchor-scan"][pl["bound_mode"]],
                         "batches": pl["batches"], "lanes_per_query": pl["lanes_per_query"], "tail_iter": st["tail_iter"]}}
        o = orc.Oracle(name, **kw)
        fo, go, _ = o.cost_function(w["points"], xs[0], nthreads=cores)
        case["cost_rel_err_vs_oracle"] = abs(fh - fo) / max(abs(fo), 1e-300)
        case["grad_rel_err_vs_oracle"] = float(np.linalg.norm(gh - go) / max(np.linalg.norm(go), 1e-300))
        for label, nt in (("oracle_ms_threads_%d" % w["threads_num"], w["threads_num"]), ("oracle_ms_all_cores", cores)):
            ts = []
            o.cost_function(w["points"], xs[0], nthreads=nt)
            for k in range(oracle_reps):
                t0 = time.perf_counter()
                o.cost_function(w["points"], xs[k % len(xs)], nthreads=nt)
                ts.append(1e3 * (time.perf_counter() - t0))
            case[label] = float(np.median(ts))
        case["host_cores"] = cores
        case["speedup_vs_oracle_threads_%d" % w["threads_num"]] = 1e3 * case["oracle_ms_threads_%d" % w["threads_num"]] / case["callback_us_median"]
        out["cases"][name] = case
        c.close()
    out["note"] = ("reference demo maps (data fixtures of src/plan_manager/pcds/map_*.pcd) -> occupancy grid -> AABB gather "
                   "around 23 waypoints (plan_manager.cpp:156-175); full callback costFunctionLmbmParallel (BEO:344-408) "
                   "through svsdf_lmbm_evaluate, host MINCO included; callback_us_* = wall time of the foreign call "
                   "(it synchronises before returning); device_us = HIP-event span of one call's device work; the "
                   "oracle is the CPU restatement (kind 'port'), OpenMP schedule(dynamic) like BEO:785")
    return out


def sustained(r, steps=300, seed=11):
    """The headline workload in the regime and for the length an optimisation runs it: `steps` (>= 300,
    about 2 s) evaluations with generic piece durations back to back, per-step wall time, and the shader clock each step
    ran at (svsdf_stats.shader_clock_mhz: measured by the kernel itself)."""
    import svsdf_amd
    rng = np.random.default_rng(seed)
    N = r.N
    x = r.x.copy()
    x[:N] *= 1.0 + 1e-3 * rng.standard_normal(N)
    T = svsdf_amd.forward_T(x[:N])
    coeffs = svsdf_amd.minco_coeffs(r.w["head_state"], r.w["tail_state"], x[N:].reshape(-1, 3), T)
    f = lambda: r.opt.addSaftyPenaOnSweptVolumeParallelTrueSDF(T, coeffs, 0.0, r.zT, r.zC)
    for _ in range(3):
        f()
    r.n_eval += 3 + steps
    r.fence()
    per, clk = np.zeros(steps), np.zeros(steps)
    t_all = time.perf_counter()
    for k in range(steps):
        t0 = time.perf_counter()
        f()
        per[k] = 1e3 * (time.perf_counter() - t0)
        clk[k] = r.ctx.stats().get("shader_clock_mhz", 0.0)
    r.fence()
    total = time.perf_counter() - t_all
    for _ in range(2):
        r.step()
    h = min(100, steps // 3)
    return {"steps": steps, "seconds": total, "ms_per_step": 1e3 * total / steps, "value": r.P_total * steps / total,
            "unit": "query-points/s", "first_%d_ms" % h: float(per[:h].mean()), "last_%d_ms" % h: float(per[-h:].mean()),
            "last_over_first": float(per[-h:].mean() / per[:h].mean()), "median_ms": float(np.median(per)),
            "first_%d_median_ms" % h: float(np.median(per[:h])), "last_%d_median_ms" % h: float(np.median(per[-h:])),
            "p99_ms": float(np.percentile(per, 99)), "max_ms": float(per.max()), "max_step": int(per.argmax()),
            "steps_over_1p5_median": int((per > 1.5 * np.median(per)).sum()),
            "shader_clock_mhz_first": float(clk[:h].mean()), "shader_clock_mhz_last": float(clk[-h:].mean()),
            "shader_clock_mhz_min": float(clk.min()), "piece_time_exact": r.ctx.stats()["piece_time_exact"],
            "note": "generic piece durations (the production regime); shader clock = s_memtime cycles per s_memrealtime "
                    "cycle over the life of the main solve's first wave, read by the kernel itself; the interpreter's objects "
                    "are frozen out of the cyclic collector (main(): gc.freeze) -- until then a full collection over "
                    "torch's ~ 1e6 objects put one 35 ms pause at a random step (profiles/r05_stall_probe.txt)"}


def stripe_report(r, devices, steps=5):
    """8-GPU readiness that can be measured on any box: every stripe's own device time (the stripes
    evaluated ONE AFTER THE OTHER with per-launch HIP events, so that stripes sharing a GPU do not stretch each other),
    the plan each stripe follows, and -- concurrently again -- what the fan-out costs the host per evaluation."""
    c = r.ctx
    G = len(devices)
    c.set_group_serial(True)
    c.set_profiling(3)     # span only: the per-launch events of level 1 stretch a 500 k-point stripe by ~ 5 % (3.85 vs 3.67 ms)
    r.step()
    dev = np.zeros((steps, G))
    for k in range(steps):
        r.step()
        for j in range(G):
            dev[k, j] = c.group_stripe(j)["stats"]["device_ms"]
    info = [c.group_stripe(j) for j in range(G)]
    c.set_profiling(True)  # and once with them, for the record
    r.step()
    r.step()
    dev_ev = np.array([c.group_stripe(j)["stats"]["device_ms"] for j in range(G)])
    c.set_profiling(False)
    c.set_group_serial(False)
    for _ in range(2):
        r.step()
    fan, comb, wall = [], [], []
    for _ in range(max(steps, 10)):
        t0 = time.perf_counter()
        r.step()
        wall.append(1e3 * (time.perf_counter() - t0))
        st = c.stats()
        fan.append(1e3 * st["fanout_ms"])
        comb.append(1e3 * st["combine_ms"])
    d = dev.mean(axis=0)
    modes = ["cheap-chunk", "full-scan", "lazy-scan", "anchor-scan"]
    return {"per_stripe": [{"stripe": j, "device": info[j]["device"], "points": info[j]["points"],
                            "interior_points": int(info[j]["stats"]["interior_points"]), "device_ms_alone": float(d[j]),
                            "plan": {"gsip_bound_mode": modes[info[j]["plan"]["bound_mode"]], "batches": info[j]["plan"]["batches"],
                                     "lanes_per_query": info[j]["plan"]["lanes_per_query"]}} for j in range(G)],
            "device_ms_max": float(d.max()), "device_ms_mean": float(d.mean()), "device_ms_max_over_mean": float(d.max() / d.mean()),
            "device_ms_mean_with_per_launch_events": float(dev_ev.mean()),
            "fanout_us_per_evaluation": float(np.median(fan)), "combine_us_per_evaluation": float(np.median(comb)),
            "fixed_host_us_per_evaluation": float(np.median(fan) + np.median(comb)),
            "ideal_ms_per_step_on_%d_gpus" % G: float(d.max() + 1e-3 * (np.median(fan) + np.median(comb))),
            "concurrent_ms_per_step_here": float(np.median(wall)),
            "note": "device_ms_alone: HIP-event span (first to last event of the evaluation, svsdf_set_profiling 3: no per-launch events) of the stripe's pipeline with the stripes run one after the other "
                    "(svsdf_set_group_serial) -- on a box with one GPU per stripe that is the stripe's time; fanout = waking "
                    "the per-device host threads + joining them, combine = fixed-order host sum of the partials; "
                    "ideal = slowest stripe + the fixed host cost"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import svsdf_amd
    from svsdf_amd import workload

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # The interpreter's cyclic collector: with torch imported a full collection walks ~ 1e6 objects -- one 35 ms pause every
    # few hundred evaluations, i.e. inside a timed region now and then (tools/stall_probe.py: the same loop on the bare
    # context has none in 1 500 steps).  Everything alive now is moved to the permanent generation; later collections only
    # see what the loops allocate.  (The reference's host is C++.)
    gc.collect()
    gc.freeze()
    if a.only == "reference_scale":
        print(json.dumps({"reference_scale": reference_scale(local_rank)}), flush=True)
        return
    if os.environ.get("SVSDF_BENCH_ONE_GPU"):     # emulation on a 1-GPU box: every rank / stripe on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    tdist = None
    backend = os.environ.get("SVSDF_BENCH_BACKEND", "nccl")   # "gloo": emulation of N ranks on one GPU
    if world > 1 or os.environ.get("SVSDF_BENCH_FORCE_DIST"):   # the latter exercises the RCCL path on 1 GPU
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            tdist.init_process_group(backend, rank=rank, world_size=world)
    devices = None
    n_gpus = world
    if world == 1 and a.gpus > 1:
        a.inprocess = True      # plain `python bench.py --gpus N`: ONE process drives N devices through the C ABI
    if a.inprocess:
        if world != 1:
            raise SystemExit("--inprocess is a single-process mode (do not launch it with torchrun)")
        devices = [int(v) for v in a.devices.split(",")] if a.devices else list(range(a.gpus))
        if not a.devices and torch.cuda.device_count() < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible; pass --devices "
                             "with repeated ordinals to put several stripes on one GPU on purpose")
        n_gpus = len(devices)
        if n_gpus < 2 and a.combine != "rccl":   # (one device + rccl: a 1-rank communicator, measures the collective's fixed cost)
            devices = None
    elif world > 1 and a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    multi = n_gpus > 1
    name = a.config or ("C4" if multi else "C3")
    P_total = a.points or workload.CONFIGS[name]["P"]

    r = Runner(a, name, P_total, a.dist, rank, world, local_rank, tdist, devices)
    w, N, ctx = r.w, r.N, r.ctx
    setup = r.settle()
    elapsed, acc, last, combine_ms = r.timed(a.steps, a.warmup)
    prof_steps = max(1, min(a.steps, 5))
    solve_ms_step, dev_ms_step = r.profiled(prof_steps)
    r.fence()
    cb_steps = max(1, min(a.steps, 10))
    full_cb_ms = r.full_callback(cb_steps)
    full_cb_pert_ms = r.full_callback(cb_steps, perturb=1e-3)
    generic = r.generic_durations(max(1, min(a.steps, 20))) if not (tdist is not None or (a.inprocess and a.gpus > 1)) else None
    shard_points = last["points"]          # points resident on this process' device(s)
    evals_rank, interior_rank = acc["sdf_evals"], last["interior_points"]
    ar_ms = None
    if tdist is not None:
        if backend == "nccl":
            ar_ms = allreduce_ms(tdist, 19 * N + 1)
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            agg = torch.tensor([acc["sdf_evals"], acc["solves"], last["interior_points"], acc["scan_evals"],
                                last["culled_points"]], dtype=torch.float64, device="cuda")
            tdist.all_reduce(agg, op=tdist.ReduceOp.SUM)
        else:
            t = torch.tensor([elapsed], dtype=torch.float64)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            agg = torch.tensor([acc["sdf_evals"], acc["solves"], last["interior_points"], acc["scan_evals"],
                                last["culled_points"]], dtype=torch.float64)
            tdist.all_reduce(agg, op=tdist.ReduceOp.SUM)
        elapsed = float(t.item())
        evals_all, solves_all, interior_all, scan_all, culled_all = [float(v) for v in agg.tolist()]
    else:
        evals_all, solves_all, interior_all, scan_all, culled_all = (float(acc["sdf_evals"]), float(acc["solves"]),
                                                                     float(last["interior_points"]), float(acc["scan_evals"]),
                                                                     float(last["culled_points"]))
    if rank != 0:
        if tdist is not None:
            tdist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / a.steps
    value = P_total * a.steps / elapsed
    # dominant kernel = k_solve: rank-0 (in-process: slowest device) HIP-event time on the library's own streams
    pts_dev = shard_points / (n_gpus if a.inprocess and multi else 1)
    ach_gbs = BYTES_PER_POINT * pts_dev / (solve_ms_step * 1e-3) / 1e9
    fp64 = fp64_accounting(acc, a.steps, solve_ms_step, r.solve_ms_serial, r.round_ms, r.round_ms_serial, ms_per_step,
                           ndev=(n_gpus if a.inprocess and multi else 1))
    traffic = None
    traffic_total = None
    try:  # PMC-derived HBM traffic of k_solve per evaluation (collected offline, see profiles/)
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = tr.get(f"{name}:{a.dist}:{int(pts_dev)}")
        traffic_total = tr.get(f"{name}:{a.dist}:{int(pts_dev)}:total")
    except Exception:
        pass
    if multi:
        how = (f"ONE process, {n_gpus} devices {devices} through the C ABI's multi-device context, "
               f"{['host', 'host', 'rccl'][last['combine']]} combine of the partials"
               if a.inprocess else f"{world} processes (torchrun), 1 GPU each, RCCL all-reduce via torch.distributed")
    else:
        how = "1 GPU, no collective"
    res = {
        "metric": "SVSDF query-points/sec (cost+grad) per optimizer evaluation",
        "value": value, "unit": "query-points/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if multi else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{name}: {w['shape']} shape, {N}-piece MINCO (2.5 s/piece), {P_total} {a.dist} "
                               f"query points in total ({int(pts_dev)} per GPU), seed {workload.SEED}",
                   "points_total": P_total, "points_per_gpu": int(pts_dev), "pieces": N, "shape": w["shape"],
                   "distribution": a.dist,
                   "interior_fraction": interior_all / P_total,
                   "culled_fraction": culled_all / P_total,
                   "argmin_solves_per_point": solves_all / a.steps / P_total,
                   "gsip_samples_per_point_rank0": acc["gsip_samples"] / a.steps / max(shard_points, 1),
                   "parallelism": f"points striped over {n_gpus} GPU(s) ({how}); one sum of {19 * N + 1} f64 per evaluation"},
        "setup": setup,
        "full_callback_ms": full_cb_ms,
        "full_callback_perturbed_ms": full_cb_pert_ms,
        "full_callback_note": "a14 = costFunctionLmbmParallel: tau -> T, host MINCO forward, penalty (this bench's step), "
                              "MINCO adjoint, chain rule; mean of %d calls.  'perturbed': every call gets x * (1 + 1e-3 * "
                              "N(0,1)), i.e. the launch plan learned from the previous call is slightly off, as in a "
                              "real optimisation" % cb_steps,
        "roofline": {"bound": "hbm", "kernel": "k_solve (argmin over t: pruned table scan + scan layers 2-4 + descent; all launches of one evaluation)",
                     "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "traffic_total": traffic_total,
                     "traffic_unit": "bytes per evaluation (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes); `traffic` = the dominant "
                                     "kernel k_solve alone, `traffic_total` = every kernel of the evaluation (k_prep ... k_finish; most of it is "
                                     "k_round's hand-off of GSIP samples through HBM)",
                     "kernel_ms_per_step": solve_ms_step, "kernel_ms_sum_per_step": r.solve_ms_sum,
                     "kernel_time_note": "a large shard runs as several point batches on concurrent streams (setup.batches): kernel_ms_per_step = time "
                                         "during which at least one k_solve launch was executing (merged HIP-event intervals; "
                                         "what `achieved` divides by), kernel_ms_sum_per_step = plain sum of the launch durations "
                                         "(what a rocprofv3 kernel trace adds up: launches x average duration)",
                     "kernel_ms_serialized_per_step": r.solve_ms_serial, "device_ms_serialized_per_step": r.dev_ms_serial,
                     "serialized_note": "the same evaluation with the batches run one after the other (svsdf_set_profiling 2; "
                                        "what `SVSDF_BATCHES=1` gives): k_solve's own cost per evaluation, comparable with a "
                                        "kernel trace taken that way (profiles/*_b1_kernel_stats.csv) and with round 1",
                     "launches_per_step": acc["solve_launches"] / a.steps,
                     "device_ms_per_step": dev_ms_step, "profiled_steps": prof_steps,
                     "note": "24 B/point algorithmic; the solve is FP64-VALU bound (SURVEY.md §8d), see fp64",
                     "k_round_ms_per_step": r.round_ms, "k_round_ms_sum_per_step": r.round_ms_sum,
                     "k_round_ms_serialized_per_step": r.round_ms_serial,
                     "fp64": fp64},
    }
    res["shader_clock_mhz"] = last.get("shader_clock_mhz", 0.0)
    if a.inprocess and multi:
        # both ways of summing the devices' partials, timed in this run on this workload
        res["combine_ab"] = combine_ab(r, a, devices, max(3, min(a.steps, 10)))
        try:
            res["stripes"] = stripe_report(r, devices)
        except Exception as e:      # (a library from before round 5 has no per-stripe entry points)
            res["stripes"] = {"error": str(e)}
    if generic is not None:
        res["generic_durations"] = generic
        if not a.no_extras and a.steps >= 5:
            res["sustained"] = sustained(r, max(300, a.sustained_steps))
    res["evaluations_in_this_run"] = r.n_eval   # of the headline workload (settle + warm-up + timed + profiled + callbacks)
    if multi or devices is not None or ar_ms is not None:
        res["combine"] = {"ms_allreduce": ar_ms, "ms_combine_inprocess": combine_ms if a.inprocess else None,
                          "mode": ["host", "host", "rccl"][last["combine"]] if a.inprocess else "torch.distributed",
                          "rccl_ranks": (r.ctx.group_info()["rccl_ranks"] if a.inprocess else (world if ar_ms is not None else 0)),
                          "note": "ms_allreduce: RCCL all-reduce of 19N+1 doubles alone (launch + sync, torchrun path); "
                                  "ms_combine_inprocess: host time from 'all devices done' to 'summed partial on the host'"}
    # release the headline workload before the sub-runs
    r.opt._ctx.close()
    if multi and not a.no_extras:
        # strong-scaling base measured in the same run: the same total cloud on ONE GPU (rank 0's / the first device),
        # so that the speed-up of this line does not have to be inferred from the N = 1 line of another workload
        import svsdf_amd
        c1 = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                    poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"],
                                    tail_state=w["tail_state"], device=(devices[0] if devices else local_rank))
        c1.set_points(w["points"])
        for _ in range(3):
            c1.eval_penalty(w["coeffs"], w["T"])
        nb = max(3, min(a.steps, 10))
        t0 = time.perf_counter()
        for _ in range(nb):
            c1.eval_penalty(w["coeffs"], w["T"])
        base_ms = 1e3 * (time.perf_counter() - t0) / nb
        c1.close()
        res["strong_scaling_base"] = {"n_gpus": 1, "points_total": P_total, "steps": nb, "ms_per_step": base_ms,
                                      "value": P_total / (base_ms * 1e-3), "speedup_of_this_line": base_ms / ms_per_step,
                                      "note": "same workload, whole cloud resident on one GPU, measured by rank 0 after the timed region"}
    if not multi and not a.no_extras and a.config is None and a.points is None and a.dist == "corridor":
        res["north_star"] = sub_run(a, "NS", workload.CONFIGS["NS"]["P"], "corridor", rank, world, local_rank, None, None, a.extra_steps)
        res["map_distribution"] = sub_run(a, name, P_total, "map", rank, world, local_rank, None, None, a.extra_steps)
    if not multi and not a.no_extras and a.config is None and a.points is None and a.dist == "corridor":
        # every other BASELINE config at its full size on this GPU (compact objects), C4's whole 4 M cloud on ONE device
        # (the base of the N > 1 lines, which run C4), and what one optimisation pays from a cold context
        res["other_configs"] = {c: quick_run(c, workload.CONFIGS[c]["P"], st_, local_rank, generic=True)
                                for c, st_ in (("C1", 20), ("C2", 20), ("C5", 5))}
        res["c4_one_gpu"] = quick_run("C4", workload.CONFIGS["C4"]["P"], 5, local_rank)
        res["first_optimisation"] = first_optimisation(name, P_total, local_rank)
        res["reference_scale"] = reference_scale(local_rank)
    if not multi and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(w, a.cpu_seconds)
        res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
        if "north_star" in res:
            wn = workload.make("NS", minco=svsdf_amd.minco_coeffs)
            cb = cpu_baseline(wn, a.cpu_seconds / 2)
            res["north_star"]["cpu_baseline"] = cb
            res["north_star"]["speedup_vs_cpu_baseline"] = res["north_star"]["value"] / cb["value"]
    if tdist is not None:
        tdist.destroy_process_group()
    # RCCL writes a version banner to the C stdio buffer at communicator creation: push it out first so that the
    # JSON line is the LAST line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
