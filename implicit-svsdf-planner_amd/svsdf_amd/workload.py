"""Deterministic synthetic workloads for the SVSDF hot path (BASELINE.json configs).

Trajectory: N MINCO pieces of `inittime` = 2.5 s from the recorded Start to End pose of the
reference's demo for that shape (src/plan_manager/pcds/trajectory_<shape>.txt), interior
waypoints on a smooth weaving SE(2) path.  Points: "corridor" (reference-like: union of boxes
around the waypoints, mirroring src/plan_manager/src/plan_manager.cpp:156-173 with half-width
kernel_size * occupancy_resolution / 3) or "map" (uniform over the demo map extent).
Seed 20240807, numpy PCG64 (identical numpy on the build and GPU boxes).
"""
import json
import os

import numpy as np

SEED = 20240807

# Data lifted from the reference's demo assets (values, not code):
#   src/plan_manager/pcds/trajectory_<shape>.txt, src/plan_manager/config/<shape>.yaml
SCENARIOS = {
    "star": dict(start=(4.3987178802490234, 4.7499313354492188), end=(20.23274040222168, 64.403488159179688),
                 safety_hor=0.7, kernel_size=17, poly_params=(0.0, 0.0, 0.0)),
    "sdHorseshoe": dict(start=(21.929414749145508, 61.368782043457031), end=(3.7336540222167969, 2.9125022888183594),
                        safety_hor=0.7, kernel_size=17, poly_params=(0.0, 0.0, 0.0)),
    "sdHeart": dict(start=(15.966060638427734, 62.657791137695312), end=(19.974319458007812, 3.9720420837402344),
                    safety_hor=0.8, kernel_size=21, poly_params=(0.0, 0.0, 0.0)),
    "sdUnevenCapsule": dict(start=(14.187297821044922, 66.808502197265625), end=(5.8466892242431641, 3.8529911041259766),
                            safety_hor=0.7, kernel_size=21, poly_params=(0.0, 0.0, 0.0)),
    "sdCutDisk": dict(start=(3.9490070343017578, 5.3206806182861328), end=(8.3104610443115234, 65.800971984863281),
                      safety_hor=0.87, kernel_size=17, poly_params=(0.0, -3.0, 0.0)),
}
WEIGHT_P = 60.0
RHO = 3.8
INITTIME = 2.5
OCC_RES = 1.0
MAP_EXTENT = ((0.0, 30.5), (0.0, 75.0))  # bounds of src/plan_manager/pcds/map_*.pcd

# BASELINE.json configs.  C5 ("arbitrary .obj mesh, no analytic shape SDF"): the reference's own mesh of the star
# (src/plan_manager/shapes/star.obj, 152 vertices / 300 triangles, committed as data in tests/golden/reference_assets.json)
# -> its z = 0 outline (77 vertices, svsdf_mesh_outline) -> the generic Polygon shape.
CONFIGS = {
    "C1": dict(shape="star", N=8, P=10_000),
    "C2": dict(shape="star", N=16, P=100_000),
    "C3": dict(shape="sdHorseshoe", N=32, P=1_000_000),
    "C4": dict(shape="sdHeart", N=32, P=4_000_000),
    "C5": dict(shape="Polygon", N=16, P=1_000_000, scenario="star", mesh="star"),
    # the workload BASELINE.json's north_star target is quoted on: "1M-query-point / 16-segment MINCO
    # cost+grad evaluation at 1 GPU" with the demo shape of configs[0..1]
    "NS": dict(shape="star", N=16, P=1_000_000),
}


MESH_NAMES = ["sdArc", "sdCutDisk", "sdHeart", "sdHorseshoe", "sdOrientedVesica", "sdPie", "sdPie2", "sdRhombus",
              "sdRoundedCross", "sdRoundedX", "sdTunnel", "sdUnevenCapsule", "star"]   # src/plan_manager/shapes/*.obj
_ASSETS = None


def reference_mesh(name):
    """(V (nv, 3), F (nf, 3) zero-based) of the reference's shapes/<name>.obj, from the committed data fixture
    (tests/golden/reference_assets.json, written by tests/golden/make_fixtures.py)."""
    a = _assets()
    return np.array(a["shapes"][name], dtype=np.float64), np.array(a["mesh_faces"][name], dtype=np.int32)


def mesh_outline(name, z0=0.0):
    """z = z0 outline of the reference mesh `name` through the product's svsdf_mesh_outline (host C++)."""
    from . import binding
    V, F = reference_mesh(name)
    xy, loops = binding.mesh_outline(V, F, z0)
    if loops != 1:
        raise ValueError(f"{name}: {loops} closed loops in the z = {z0} section")
    return xy


def star_outline():
    """10-vertex outline of the reference's analytic `star` (r = 2.8, rf = 0.6; SHP:565-566): a small hand-made
    Polygon for unit tests (config C5 uses the 77-vertex outline of the star MESH, mesh_outline("star"))."""
    r, rf = 2.8, 0.6
    # iq's sdStar5: outer tips at radius r, inner vertices where the two mirrored edges meet
    k1 = np.array([0.809016994375, -0.587785252292])
    ba = rf * np.array([-k1[1], k1[0]]) - np.array([0.0, 1.0])
    inner = np.array([0.0, r]) + ba * r  # end of the clipped segment (h = r)
    ri = np.hypot(*inner)
    pts = []
    for i in range(5):
        a = np.pi / 2 + 2 * np.pi * i / 5
        pts.append((r * np.cos(a), r * np.sin(a)))
        b = a + np.pi / 5
        pts.append((ri * np.cos(b), ri * np.sin(b)))
    return np.array(pts)


def waypoints(start, end, N, amp=3.0):
    """Interior waypoints q_i (N-1, 3) = (x, y, yaw)."""
    s = np.array([start[0], start[1], 0.0])
    e = np.array([end[0], end[1], 0.0])
    u = (np.arange(N - 1) + 1.0) / N
    q = s[None, :] * (1.0 - u[:, None]) + e[None, :] * u[:, None]
    w = np.sin(np.pi * u)  # offsets vanish at both ends (head/tail states are at rest)
    q[:, 0] += w * amp * np.sin(2 * np.pi * 3 * u)
    q[:, 1] += w * amp * np.cos(2 * np.pi * 2 * u)
    q[:, 2] = 0.8 * np.sin(2 * np.pi * 1.5 * u)
    return q


def states(start, end):
    hs = np.zeros((3, 3))
    ts = np.zeros((3, 3))
    hs[:2, 0] = start
    ts[:2, 0] = end
    return hs, ts


def corridor_points(q, P, half, rng):
    """Uniform samples of the union of boxes max-norm(p - q_i) <= half (rejection keeps the
    density uniform where boxes overlap)."""
    out = np.zeros((P, 3))
    n = 0
    c = q[:, :2]
    while n < P:
        m = max(1024, int((P - n) * 2.5))
        bi = rng.integers(0, len(c), m)
        p = c[bi] + rng.uniform(-half, half, (m, 2))
        # multiplicity = number of boxes containing p
        mult = np.zeros(m, dtype=np.int64)
        for j in range(len(c)):
            mult += (np.max(np.abs(p - c[j]), axis=1) <= half)
        keep = rng.uniform(0.0, 1.0, m) * mult < 1.0
        p = p[keep]
        k = min(len(p), P - n)
        out[n:n + k, :2] = p[:k]
        n += k
    return out


def map_points(P, rng):
    out = np.zeros((P, 3))
    out[:, 0] = rng.uniform(*MAP_EXTENT[0], P)
    out[:, 1] = rng.uniform(*MAP_EXTENT[1], P)
    return out


def make(config="C2", P=None, N=None, dist="corridor", seed=SEED, minco=None):
    """Returns a dict: shape, polygon, safety_hor, weight_p, rho, poly_params, head_state,
    tail_state, q (N-1, 3), T (N,), coeffs (6N, 3) [if `minco` callable given], points (P, 3),
    x (the optimizer variable [tau, q])."""
    cfg = dict(CONFIGS[config]) if isinstance(config, str) else dict(config)
    shape = cfg["shape"]
    sc = SCENARIOS[cfg.get("scenario", shape)]
    N = int(N or cfg["N"])
    P = int(P or cfg["P"])
    rng = np.random.default_rng(seed)
    q = waypoints(sc["start"], sc["end"], N)
    hs, ts = states(sc["start"], sc["end"])
    T = np.full(N, INITTIME)
    if dist == "corridor":
        pts = corridor_points(q, P, sc["kernel_size"] * OCC_RES / 3.0, rng)
    elif dist == "map":
        pts = map_points(P, rng)
    else:
        raise ValueError(dist)
    w = dict(name=config if isinstance(config, str) else "custom", shape=shape, N=N, P=P, dist=dist,
             polygon=(mesh_outline(cfg["mesh"]) if cfg.get("mesh") else star_outline()) if shape == "Polygon" else None,
             safety_hor=sc["safety_hor"], weight_p=WEIGHT_P, rho=RHO, poly_params=sc["poly_params"],
             head_state=hs, tail_state=ts, q=q, T=T, points=pts)
    if minco is not None:
        w["coeffs"] = minco(hs, ts, q, T)
    return w


def _assets():
    global _ASSETS
    if _ASSETS is None:
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        _ASSETS = json.load(open(os.path.join(root, "tests", "golden", "reference_assets.json")))
    return _ASSETS


def reference_case(name, N=24, iterates=8, seed=SEED):
    """The regime the reference runs (plan_manager.cpp:156-175, back_end_optimizer.cpp:29-36): its own demo map
    (src/plan_manager/pcds/map_<name>.pcd, 210 ... 380 points; data fixture) through the query-point producer (occupancy
    grid -> AABB gather of half width kernel_size * resolution / 3 around the waypoints), N MINCO pieces from the demo's
    start to its end pose, and `iterates` optimiser variables x = [tau, q] whose durations are GENERIC doubles (2.5 s *
    (1 + 1e-3 N(0, 1)), waypoints moved by ~1 cm from one to the next) like the successive LMBM iterates of one
    optimisation.  Needs the built library (host-side producer)."""
    from . import binding
    a = _assets()
    sc = a["scenarios"][name]
    cloud = np.array(a["maps"][name], dtype=np.float32)
    q = waypoints(sc["start"][:2], sc["end"][:2], N)
    halfbd = np.full(3, sc["kernel_size"] * sc["occupancy_resolution"] / 3.0)
    pts = binding.OccupancyMap(cloud, sc["occupancy_resolution"], 1).gather(q, halfbd)
    hs, ts = states(sc["start"][:2], sc["end"][:2])
    rng = np.random.default_rng(seed)
    xs = []
    for _ in range(iterates):
        T = sc["inittime"] * (1.0 + 1e-3 * rng.standard_normal(N))
        qq = q + 1e-2 * rng.standard_normal(q.shape)
        xs.append(x_from(qq, T, binding.backward_T))
    return dict(name=name, shape=name, N=N, points=pts, map_points=len(cloud), xs=xs, q=q,
                safety_hor=sc["safety_hor"], weight_p=sc["weight_p"], rho=sc["rho"], poly_params=sc["poly_params"],
                threads_num=int(sc["threads_num"]), head_state=hs, tail_state=ts, polygon=None)


def x_from(q, T, backward_T):
    """Optimizer variable x = [tau (N), q_0 (x, y, yaw), ...] (BEO layout, a14)."""
    return np.concatenate([backward_T(T), np.asarray(q, dtype=np.float64).ravel()])
