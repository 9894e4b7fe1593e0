"""N > 1 path on CPU: world_size-2 `gloo` ranks run the product's sharding plan and the host
halves of the full callback (svsdf_lmbm_prepare / svsdf_lmbm_finish: tau->T, MINCO forward and
adjoint, partial layout) around ONE all-reduce of the (19N+1)-double partial.  The device stage
(the HIP kernels) is stood in for by the CPU oracle on each rank's shard -- here the oracle is
the checker's stand-in for the kernels, nothing of it ships."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, P, out):
    for p in (ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import svsdf_amd
    from svsdf_amd import workload
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = workload.make("C1", P=P, minco=svsdf_amd.minco_coeffs)
    N = len(w["T"])
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T) + 0.03 * np.random.default_rng(1).standard_normal(4 * N - 3)
    ctx = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"],
                                 rho=w["rho"], head_state=w["head_state"], tail_state=w["tail_state"],
                                 rank=rank, world_size=world, flags=svsdf_amd.FLAG_HOST_ONLY)
    with pytest.raises(svsdf_amd.SvsdfError):      # no device entry point works on a host-only context
        ctx.set_points(w["points"])
    coeffs, T = ctx.lmbm_prepare(x)
    mine = svsdf_amd.shard_plan(w["points"], rank, world)
    # stand-in for the device stage: this rank's partial [cost, gradC (col-major), gradT]
    o = orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                   head_state=w["head_state"], tail_state=w["tail_state"])
    o.set_traj(coeffs, T)
    c, gT, gC = o.penalty(w["points"][mine], nthreads=2)
    partial = torch.from_numpy(np.concatenate([[c], gC.T.ravel(), gT]))
    assert partial.numel() == 19 * N + 1
    dist.all_reduce(partial, op=dist.ReduceOp.SUM)
    f, g = ctx.lmbm_finish(partial.numpy(), len(x))
    if rank == 0:
        fo, go, c3 = o.cost_function(w["points"], x, nthreads=2)
        np.savez(out, f=f, g=g, fo=fo, go=go, costs=ctx.last_costs(), c3=c3, n_mine=len(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process_oracle(built, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), 600, out), nprocs=2, join=True)
    r = np.load(out)
    assert abs(r["f"] - r["fo"]) <= 1e-9 * abs(r["fo"])
    assert np.linalg.norm(r["g"] - r["go"]) <= 1e-9 * np.linalg.norm(r["go"])
    np.testing.assert_allclose(r["costs"], r["c3"], rtol=1e-9)
    assert r["n_mine"] == 300


def test_shard_plan_is_a_striped_partition(built):
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C1", P=1001)
    for world in (1, 2, 3, 8):
        parts = [svsdf_amd.shard_plan(w["points"], r, world) for r in range(world)]
        allidx = np.concatenate(parts)
        assert sorted(allidx.tolist()) == list(range(1001))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        # striping: consecutive elements of the global Morton order go to consecutive ranks, so
        # every rank's bounding box covers (almost) the whole cloud
        full = np.ptp(w["points"][:, :2], axis=0)
        for p in parts:
            assert (np.ptp(w["points"][p][:, :2], axis=0) > 0.8 * full).all()
    # Morton order keeps device neighbours spatially close
    order = svsdf_amd.shard_plan(w["points"], 0, 1)
    d_sorted = np.linalg.norm(np.diff(w["points"][order][:, :2], axis=0), axis=1).mean()
    d_input = np.linalg.norm(np.diff(w["points"][:, :2], axis=0), axis=1).mean()
    assert d_sorted < 0.2 * d_input
    keep = svsdf_amd.shard_plan(w["points"], 0, 1, flags=svsdf_amd.FLAG_KEEP_INPUT_ORDER)
    assert np.array_equal(keep, np.arange(1001))
