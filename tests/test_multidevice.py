"""In-process multi-GPU context (svsdf_config::n_devices / devices).

CPU: the combine logic -- stripe plan of a G-device context (rank * G + k of world * G), the fixed-order
host sum (svsdf_sum_partials) and the host half of the callback -- with the device stage stood in for by the
oracle on every stripe (the oracle is the checker's stand-in for the kernels, nothing of it ships).
GPU: a G-device context equals the 1-device context (on a 1-GPU box the G stripes share device 0; with more
GPUs visible they spread), host and RCCL combine, and the bench's multi-GPU modes on the BASELINE C4 workload.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NT = min(os.cpu_count() or 1, 16)


def _oracle(w):
    return orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                      poly_params=w["poly_params"], polygon=w["polygon"],
                      head_state=w["head_state"], tail_state=w["tail_state"])


@pytest.mark.parametrize("G,world", [(2, 1), (4, 1), (2, 2)])
def test_group_stripes_and_host_combine_cpu(built, G, world):
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C1", P=1200, minco=svsdf_amd.minco_coeffs)
    N = len(w["T"])
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T) + 0.02 * np.random.default_rng(3).standard_normal(4 * N - 3)
    host = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                                  head_state=w["head_state"], tail_state=w["tail_state"],
                                  flags=svsdf_amd.FLAG_HOST_ONLY)
    coeffs, T = host.lmbm_prepare(x)
    o = _oracle(w)
    o.set_traj(coeffs, T)
    seen = []
    node_partials = []
    for rank in range(world):                 # every process of the job ...
        rows = []
        for k in range(G):                    # ... drives G devices: stripe rank * G + k of world * G
            mine = svsdf_amd.shard_plan(w["points"], rank * G + k, world * G)
            seen.append(mine)
            c, gT, gC = o.penalty(w["points"][mine], nthreads=NT)
            rows.append(np.concatenate([[c], gC.T.ravel(), gT]))
        node_partials.append(svsdf_amd.sum_partials(np.array(rows)))      # the context's host combine
    allidx = np.concatenate(seen)
    assert len(allidx) == len(w["points"]) and len(np.unique(allidx)) == len(allidx)   # a partition
    sizes = [len(s) for s in seen]
    assert max(sizes) - min(sizes) <= 1
    total = svsdf_amd.sum_partials(np.array(node_partials))               # the inter-process all-reduce
    f, g = host.lmbm_finish(total, len(x))
    fo, go, _ = o.cost_function(w["points"], x, nthreads=NT)
    assert abs(f - fo) <= 1e-11 * abs(fo)
    np.testing.assert_allclose(g, go, rtol=1e-9, atol=1e-9)


def test_sum_partials_is_index_order(built):
    import svsdf_amd
    rng = np.random.default_rng(0)
    p = rng.standard_normal((8, 609)) * 10.0 ** rng.integers(-8, 8, (8, 609))
    ref = p[0].copy()
    for k in range(1, 8):
        ref = ref + p[k]
    np.testing.assert_array_equal(svsdf_amd.sum_partials(p), ref)


def test_bad_device_lists_fail_cleanly(built):
    import ctypes as C
    import svsdf_amd
    from svsdf_amd import binding
    L = svsdf_amd.lib()
    cfg = binding.Config()
    L.svsdf_config_default(C.byref(cfg))
    cfg.n_devices = 9
    assert not L.svsdf_create(C.byref(cfg))
    assert b"n_devices" in L.svsdf_last_error_string(None)


# ---------------------------------------------------------------------------------------------------- GPU
def _devices(G):
    import torch
    n = torch.cuda.device_count()
    return [k % n for k in range(G)]


def _ctx(w, **kw):
    import svsdf_amd
    c = svsdf_amd.SvsdfContext(shape=w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"],
                               poly_params=w["poly_params"], polygon=w["polygon"],
                               head_state=w["head_state"], tail_state=w["tail_state"], **kw)
    c.set_points(w["points"])
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("config,P,G", [("C2", 20000, 2), ("C4", 30000, 3), ("C3", 16000, 8)])
def test_group_context_equals_single_device(built, config, P, G):
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make(config, P=P, minco=svsdf_amd.minco_coeffs)
    N = len(w["T"])
    x = workload.x_from(w["q"], w["T"], svsdf_amd.backward_T)
    one = _ctx(w, device=0)
    grp = _ctx(w, devices=_devices(G))
    assert grp.num_points() == one.num_points() == P
    for _ in range(2):                      # second pass: bound mode decided, launch plan on record
        c1, gT1, gC1 = one.eval_penalty(w["coeffs"], w["T"])
        cg, gTg, gCg = grp.eval_penalty(w["coeffs"], w["T"])
        assert abs(cg - c1) <= 1e-12 * abs(c1)
        np.testing.assert_allclose(gCg, gC1, rtol=0, atol=1e-12 * np.abs(gC1).max())
        np.testing.assert_allclose(gTg, gT1, rtol=0, atol=1e-12 * np.abs(gT1).max())
    st = grp.stats()
    assert st["n_devices"] == G and st["points"] == P and st["combine"] == svsdf_amd.COMBINE_HOST
    assert st["interior_points"] == one.stats()["interior_points"]
    f1, g1 = one.lmbm_evaluate(x)
    fg, gg = grp.lmbm_evaluate(x)
    assert abs(fg - f1) <= 1e-12 * abs(f1)
    np.testing.assert_allclose(gg, g1, rtol=0, atol=1e-12 * np.abs(g1).max())
    np.testing.assert_allclose(grp.last_costs(), one.last_costs(), rtol=1e-12)
    # per-point outputs: every stripe reports its own points; together they are the single-device answer, bit for bit
    s1, t1, q1, i1 = one.query_points(w["coeffs"], w["T"])
    sg, tg, qg, ig = grp.query_points(w["coeffs"], w["T"])
    np.testing.assert_array_equal(ig, i1)
    np.testing.assert_array_equal(sg, s1)
    np.testing.assert_array_equal(tg, t1)
    np.testing.assert_array_equal(qg, q1)


@pytest.mark.gpu
def test_group_context_empty_stripes(built):
    """Fewer points than devices: the empty stripes contribute zero and nothing hangs."""
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C1", P=3, minco=svsdf_amd.minco_coeffs)
    one = _ctx(w, device=0)
    grp = _ctx(w, devices=_devices(4))
    c1, gT1, gC1 = one.eval_penalty(w["coeffs"], w["T"])
    cg, gTg, gCg = grp.eval_penalty(w["coeffs"], w["T"])
    assert abs(cg - c1) <= 1e-12 * max(abs(c1), 1e-300)
    np.testing.assert_allclose(gCg, gC1, rtol=0, atol=1e-12 * max(np.abs(gC1).max(), 1e-300))


@pytest.mark.gpu
def test_rccl_combine(built):
    """SVSDF_COMBINE_RCCL: ncclCommInitAll + ncclAllReduce inside the library.  One rank per DISTINCT device: with
    one GPU visible this is a 1-rank communicator (still the real RCCL launch + sync path), with more it spans them."""
    import torch
    import svsdf_amd
    from svsdf_amd import workload
    w = workload.make("C2", P=12000, minco=svsdf_amd.minco_coeffs)
    n = min(torch.cuda.device_count(), 8)
    one = _ctx(w, device=0)
    grp = _ctx(w, devices=list(range(n)), combine=svsdf_amd.COMBINE_RCCL)
    c1, gT1, gC1 = one.eval_penalty(w["coeffs"], w["T"])
    cg, gTg, gCg = grp.eval_penalty(w["coeffs"], w["T"])
    assert grp.stats()["combine"] == svsdf_amd.COMBINE_RCCL and grp.stats()["n_devices"] == n
    assert abs(cg - c1) <= 1e-12 * abs(c1)
    np.testing.assert_allclose(gCg, gC1, rtol=0, atol=1e-12 * np.abs(gC1).max())
    np.testing.assert_allclose(gTg, gT1, rtol=0, atol=1e-12 * np.abs(gT1).max())
    with pytest.raises(svsdf_amd.SvsdfError, match="distinct"):
        _ctx(w, devices=[0, 0], combine=svsdf_amd.COMBINE_RCCL)
    # a stripe that fails (here: every stripe, on a non-positive duration) still JOINS the collective with a poisoned
    # partial instead of returning early and leaving the other device threads blocked in the all-reduce: the
    # call comes back with an error, and the context keeps working afterwards
    Tbad = np.array(w["T"], dtype=float)
    Tbad[3] = -1.0
    with pytest.raises(svsdf_amd.SvsdfError):
        grp.eval_penalty(w["coeffs"], Tbad)
    cg2, _, gCg2 = grp.eval_penalty(w["coeffs"], w["T"])
    assert cg2 == cg and np.array_equal(gCg2, gCg)
    st = grp.stats()
    assert st["plan_settled"] in (0, 1) and st["batches"] >= 1 and st["round_scan_evals"] >= 0


def _bench(args, env=None, launcher=None):
    e = dict(os.environ)
    e.update(env or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


@pytest.mark.gpu
def test_bench_c4_inprocess_two_stripes(built):
    """bench.py --gpus 2 --config C4 through the in-process multi-device context (both stripes on device 0 when
    only one GPU is visible): BASELINE configs[3], 4 M points in total, strong scaling."""
    import torch
    dev = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    r = _bench(["--gpus", "2", "--config", "C4", "--inprocess", "--devices", dev, "--steps", "2", "--warmup", "1"])
    assert r["n_gpus"] == 2 and r["config"]["points_total"] == 4000000 and r["scaling"] == "strong"
    assert r["config"]["points_per_gpu"] == 2000000 and r["value"] > 0
    assert r["combine"]["ms_combine_inprocess"] is not None and r["combine"]["ms_combine_inprocess"] < 1.0
    assert r["strong_scaling_base"]["points_total"] == 4000000 and r["strong_scaling_base"]["ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_gpus_n_drives_n_devices_by_itself(built):
    """A plain `python bench.py --gpus 2` (no torchrun, no --inprocess: the shape of the driver's N = 1 command with
    another N) must drive two devices through the in-process multi-device context -- or refuse, never fall back to one
    GPU and print n_gpus = 1."""
    import torch
    if torch.cuda.device_count() >= 2:
        r = _bench(["--gpus", "2", "--points", "400000", "--steps", "2", "--warmup", "1", "--no-extras"])
    else:
        e = dict(os.environ)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=e,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode != 0 and b"GPU(s) visible" in out.stderr, out.stderr.decode()[-500:]
        r = _bench(["--gpus", "2", "--devices", "0,0", "--points", "400000", "--steps", "2", "--warmup", "1", "--no-extras"])
    assert r["n_gpus"] == 2 and r["config"]["points_per_gpu"] == 200000 and r["scaling"] == "strong"
    assert r["config"]["shape"] == "sdHeart" and r["combine"]["ms_combine_inprocess"] is not None
    # both combines are timed in the same run; the RCCL communicator must hold one rank per GPU -- read
    # back from the communicator itself -- or the line must say why RCCL was not run (two stripes on one GPU)
    ab = r["combine_ab"]
    assert ab["host_ms_per_step"] > 0
    if torch.cuda.device_count() >= 2:
        assert ab["rccl_ranks"] == r["n_gpus"] == 2 and ab["rccl_ms_per_step"] > 0
    else:
        assert ab["rccl_ranks"] == 0 and ab["rccl_ms_per_step"] is None and "one rank per distinct GPU" in ab["rccl_note"]


@pytest.mark.gpu
def test_bench_one_device_rccl_reports_its_rank_count(built):
    """The RCCL combine as a 1-rank communicator (all a 1-GPU box can form): the bench line carries the rank count the
    communicator itself reports (ncclCommCount), so an N-GPU line can be checked against n_gpus."""
    r = _bench(["--gpus", "1", "--inprocess", "--combine", "rccl", "--config", "C2", "--steps", "2", "--warmup", "1",
                "--no-extras", "--no-cpu-baseline"])
    assert r["combine"]["mode"] == "rccl" and r["combine"]["rccl_ranks"] == 1


@pytest.mark.gpu
def test_bench_c4_two_ranks_emulated(built):
    """bench.py --gpus 2 --config C4 as two torchrun ranks.  With two GPUs: the real RCCL path.  With one GPU both
    ranks share device 0 and the 5 KB all-reduce goes through gloo (RCCL refuses two ranks on one device)."""
    import socket
    import torch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {} if torch.cuda.device_count() >= 2 else {"SVSDF_BENCH_ONE_GPU": "1", "SVSDF_BENCH_BACKEND": "gloo"}
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port)]
    r = _bench(["--gpus", "2", "--config", "C4", "--points", "400000", "--steps", "2", "--warmup", "1"], env, launcher)
    assert r["n_gpus"] == 2 and r["config"]["points_total"] == 400000 and r["config"]["points_per_gpu"] == 200000
    assert r["scaling"] == "strong" and r["value"] > 0
    assert r["strong_scaling_base"]["speedup_of_this_line"] > 0
