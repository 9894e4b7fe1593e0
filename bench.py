#!/usr/bin/env python3
"""bench.py -- SVSDF query-points/sec (cost+grad) per optimizer evaluation on MI355X.

One "step" = one call of the inner operator addSaftyPenaOnSweptVolumeParallelTrueSDF (reference
BEO:774-869) over the whole query-point cloud, points already resident in HBM, through the C ABI
(include/svsdf_c.h).  Default workload = BASELINE.json configs[1] (C2: star, 16-piece MINCO,
100k corridor query points per GPU; weak scaling: every rank adds another 100k points).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
FP64_PEAK_TFLOPS = 78.6    # FP64 vector peak = 1/2 of the 157.3 TF FP32 vector figure
BYTES_PER_POINT = 24.0     # algorithmic bytes per query point per evaluation (3 x f64)
FLOP_PER_EVAL = 150.0      # nominal FP64 flop per SDF-at-time evaluation


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", help="workload config of BASELINE.json (C1..C5)")
    ap.add_argument("--points", type=int, default=None, help="query points PER GPU (default: the config's)")
    ap.add_argument("--dist", default="corridor", choices=["corridor", "map"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline budget (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(w, budget_s):
    """Oracle (CPU port of the reference path, OpenMP schedule(dynamic) over points) timed on the
    host cores on a bounded prefix sample of the same workload."""
    from oracle import orc
    cores = os.cpu_count() or 1
    o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                   poly_params=w["poly_params"], polygon=w["polygon"],
                   head_state=w["head_state"], tail_state=w["tail_state"])
    o.set_traj(w["coeffs"], w["T"])
    pts = w["points"]
    probe = min(len(pts), max(256, 8 * cores))
    t0 = time.perf_counter()
    o.penalty(pts[:probe], nthreads=cores)
    tp = time.perf_counter() - t0
    n = int(min(len(pts), max(probe, probe * budget_s / max(tp, 1e-6))))
    t0 = time.perf_counter()
    o.penalty(pts[:n], nthreads=cores)
    t = time.perf_counter() - t0
    cnt = o.counters()
    return {"value": n / t, "unit": "query-points/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of {len(pts)} query points of the same workload, oracle/liborc.so "
                      f"(-O2 -fopenmp, schedule(dynamic)), {t:.2f} s",
            "sdf_evals_per_point": cnt["sdf_evals"] / max(n, 1)}


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
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("SVSDF_BENCH_FORCE_DIST"):   # the latter exercises the RCCL path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    per_gpu = a.points or workload.CONFIGS[a.config]["P"]
    P_total = per_gpu * world
    w = workload.make(a.config, P=P_total, dist=a.dist, minco=svsdf_amd.minco_coeffs)
    N = len(w["T"])

    opt = svsdf_amd.TrajOptimizer()
    opt.setParam(dict(rho=w["rho"], weight_p=w["weight_p"], safety_hor=w["safety_hor"],
                      inputdata=f"shapes/{w['shape']}.obj", poly_params=w["poly_params"],
                      polygon=w["polygon"], device=local_rank))
    opt.setConditions(w["head_state"], w["tail_state"], N)
    opt.setPoints(w["points"])          # uploaded once; each rank keeps its stripe in HBM
    ctx = opt._context()

    def step():
        return opt.addSaftyPenaOnSweptVolumeParallelTrueSDF(w["T"], w["coeffs"], 0.0, np.zeros(N), np.zeros((6 * N, 3)))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(6):      # part of the setup: the library settles its launch plan (lane-group widths, GSIP bound
        step()              # mode) over the first six evaluations after set_points; all give identical results
    for _ in range(a.warmup):
        step()
    fence()
    evals = scan = solves = launches = interior = samples = culled = 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
        st = ctx.stats()
        evals += st["sdf_evals"]; scan += st["scan_evals"]; solves += st["solves"]
        launches += st["solve_launches"]; interior = st["interior_points"]; samples += st["gsip_samples"]; culled = st["culled_points"]
    fence()
    elapsed = time.perf_counter() - t0
    # kernel times of the dominant kernel: separate passes with per-launch HIP events on the
    # library's streams (event records add ~6 us per launch, so they stay out of the timed region)
    prof_steps = max(1, min(a.steps, 5))
    ctx.set_profiling(True)
    solve_ms = dev_ms = 0.0
    for _ in range(prof_steps):
        step()
        st = ctx.stats()
        solve_ms += st["solve_ms"]; dev_ms += st["device_ms"]
    ctx.set_profiling(False)
    fence()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([evals, solves, interior * a.steps, scan], dtype=torch.float64, device="cuda")
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        evals_all, solves_all, interior_all, scan_all = [float(v) for v in agg.tolist()]
    else:
        evals_all, solves_all, interior_all, scan_all = float(evals), float(solves), float(interior * a.steps), float(scan)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    ms_per_step = 1e3 * elapsed / a.steps
    value = P_total * a.steps / elapsed
    # dominant kernel = k_solve; rank-0 HIP-event time on the library's own streams
    solve_ms_step = solve_ms / prof_steps
    shard = ctx.num_points()
    ach_gbs = BYTES_PER_POINT * shard / (solve_ms_step * 1e-3) / 1e9
    ach_tf = (evals / a.steps) * FLOP_PER_EVAL / (solve_ms_step * 1e-3) / 1e12
    traffic = None
    try:  # PMC-derived HBM traffic of k_refine per evaluation (collected offline, see profiles/)
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = tr.get(f"{a.config}:{a.dist}:{per_gpu}")
    except Exception:
        pass
    res = {
        "metric": "SVSDF query-points/sec (cost+grad) per optimizer evaluation",
        "value": value, "unit": "query-points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{a.config}: {w['shape']} shape, {N}-piece MINCO (2.5 s/piece), "
                               f"{per_gpu} {a.dist} query points per GPU, seed {workload.SEED}",
                   "points_total": P_total, "pieces": N, "shape": w["shape"], "distribution": a.dist,
                   "interior_fraction": interior_all / a.steps / P_total,
                   "argmin_solves_per_point": solves_all / a.steps / P_total,
                   "gsip_samples_per_point_rank0": samples / a.steps / max(ctx.num_points(), 1),
                   "culled_fraction_rank0": culled / max(ctx.num_points(), 1),
                   "parallelism": f"points striped over {world} GPU(s), 1 all-reduce of {19 * N + 1} f64"},
        "roofline": {"bound": "hbm", "kernel": "k_solve (argmin over t: pruned table scan + scan layers 2-4 + descent; all launches of one evaluation)",
                     "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "traffic_unit": "bytes per evaluation (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                     "kernel_ms_per_step": solve_ms_step, "launches_per_step": launches / a.steps,
                     "device_ms_per_step": dev_ms / prof_steps, "profiled_steps": prof_steps,
                     "note": "24 B/point algorithmic; the solve is FP64-VALU bound (SURVEY.md §8d), see fp64",
                     "fp64": {"bound": "fp64_valu", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach_tf / FP64_PEAK_TFLOPS,
                              "sdf_evals_per_step": evals / a.steps, "layer1_evals_per_step": scan / a.steps,
                              "flop_per_eval_nominal": FLOP_PER_EVAL}},
    }
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(w, a.cpu_seconds)
        res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
