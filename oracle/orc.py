"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SHAPES = ["sdUnevenCapsule", "sdCutDisk", "sdTrapezoid", "sdRhombus", "star", "sdTunnel",
          "sdHorseshoe", "sdHeart", "sdOrientedVesica", "sdRoundedCross", "sdRoundedX", "bigX",
          "sdMoon", "sdPie", "sdPie2", "sdArc", "Polygon"]
SHAPE_ID = {n: i for i, n in enumerate(SHAPES)}

_dp = C.POINTER(C.c_double)


class Counters(C.Structure):
    _fields_ = [("sdf_evals", C.c_longlong), ("shape_evals", C.c_longlong),
                ("solves", C.c_longlong), ("interior_points", C.c_longlong),
                ("gd_trials", C.c_longlong), ("gd_passes", C.c_longlong), ("gd_max_passes", C.c_longlong),
                ("gd_pass_hist", C.c_longlong * 32)]


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "svsdf_oracle.c")
    hdr = os.path.join(_HERE, "svsdf_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(so)
        for f in (src, hdr, os.path.join(_HERE, "Makefile")))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_set_traj.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.orc_traj_duration.restype = C.c_double
    L.orc_traj_duration.argtypes = [C.c_void_p]
    L.orc_traj_pos.argtypes = [C.c_void_p, C.c_double, _dp]
    L.orc_traj_vel.argtypes = [C.c_void_p, C.c_double, _dp]
    L.orc_sdf_at_time.restype = C.c_double
    L.orc_sdf_at_time.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    L.orc_sdf_swept.restype = C.c_double
    L.orc_shape_eval_batch.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp]
    L.orc_shape_eval_batch.restype = None
    L.orc_sdf_swept.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp, _dp]
    L.orc_true_sdf.restype = C.c_double
    L.orc_true_sdf.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp, _dp]
    L.orc_query.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_int, _dp, _dp, _dp]
    L.orc_penalty.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    L.orc_set_trig_mode.argtypes = [C.c_void_p, C.c_int]
    L.orc_set_modes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_set_trig_perturb.argtypes = [C.c_void_p, C.c_ulonglong]
    L.orc_cost_function.restype = C.c_double
    L.orc_cost_function.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_int, _dp, _dp, C.c_int, _dp]
    L.orc_minco_coeffs.argtypes = [_dp, _dp, C.c_int, _dp, _dp, _dp]
    L.orc_forward_T.argtypes = [_dp, _dp, C.c_int]
    L.orc_backward_T.argtypes = [_dp, _dp, C.c_int]
    L.orc_smoothed_l1.restype = C.c_int
    L.orc_smoothed_l1.argtypes = [C.c_double, C.c_double, _dp, _dp]
    L.orc_shape_id_from_name.restype = C.c_int
    L.orc_shape_id_from_name.argtypes = [C.c_char_p]
    L.orc_map_points.restype = C.c_size_t
    L.orc_map_points.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_double, C.c_int, _dp, C.c_size_t, _dp, _dp,
                                 C.c_size_t, C.POINTER(C.c_int)]
    _u8p = C.POINTER(C.c_ubyte)
    L.orc_check_sub_sw_collision.restype = C.c_int
    L.orc_check_sub_sw_collision.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_size_t]
    L.orc_shape_kernels.restype = C.c_int
    L.orc_shape_kernels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, _u8p, _u8p, _dp]
    _LIB = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One reference-equivalent (TrajOptimizer + SweptVolumeManager) state on the CPU."""

    def __init__(self, shape, safety_hor=0.7, weight_p=60.0, rho=3.8, poly_params=(0.0, 0.0, 0.0),
                 polygon=None, head_state=None, tail_state=None, polygon_loops=None):
        """polygon_loops: vertex counts of the closed loops `polygon` is made of (None: one loop, the reference's chain)."""
        self.L = lib()
        sid = SHAPE_ID[shape] if isinstance(shape, str) else int(shape)
        pp = _f64(poly_params)
        poly = None if polygon is None else _f64(polygon).reshape(-1, 2)
        hs = _f64(np.zeros((3, 3)) if head_state is None else head_state)
        ts = _f64(np.zeros((3, 3)) if tail_state is None else tail_state)
        # 3x3 states are passed column-major (col0=pos, col1=vel, col2=acc)
        self._hs = np.asfortranarray(hs).ravel(order="F").copy()
        self._ts = np.asfortranarray(ts).ravel(order="F").copy()
        self.ctx = C.c_void_p(self.L.orc_create(sid, _p(pp), _p(poly), 0 if poly is None else len(poly),
                                                safety_hor, weight_p, rho, _p(self._hs), _p(self._ts)))
        if polygon_loops is not None and len(polygon_loops) > 1:
            ls = (C.c_int * len(polygon_loops))(*[int(v) for v in polygon_loops])
            self.L.orc_set_polygon_loops.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
            if self.L.orc_set_polygon_loops(self.ctx, ls, len(polygon_loops)):
                raise ValueError("polygon_loops do not fit the vertex list")
        self.N = 0

    def __del__(self):
        try:
            if self.ctx:
                self.L.orc_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # coeffs: (6N, 3) array (row 6i+k = coefficient of s^k of piece i); T: (N,)
    def set_traj(self, coeffs, T):
        T = _f64(T)
        cm = np.asfortranarray(_f64(coeffs)).ravel(order="F").copy()
        self.N = len(T)
        self.L.orc_set_traj(self.ctx, self.N, _p(cm), _p(T))

    def duration(self):
        return self.L.orc_traj_duration(self.ctx)

    def pos(self, t):
        o = np.zeros(3)
        self.L.orc_traj_pos(self.ctx, t, _p(o))
        return o

    def vel(self, t):
        o = np.zeros(3)
        self.L.orc_traj_vel(self.ctx, t, _p(o))
        return o

    def sdf_at_time(self, px, py, t):
        return self.L.orc_sdf_at_time(self.ctx, px, py, t)

    def shape_eval(self, xy, grad=False):
        """Raw body-frame getonlySDF (and getonlyGrad1) of this oracle's shape at (P, 2) points."""
        xy = _f64(xy).reshape(-1, 2)
        sdf = np.zeros(len(xy))
        g = np.zeros((len(xy), 2)) if grad else None
        self.L.orc_shape_eval_batch(self.ctx, _p(xy), len(xy), _p(sdf), _p(g))
        return (sdf, g) if grad else sdf

    def sdf_swept(self, px, py):
        t = C.c_double(0.0)
        g = np.zeros(3)
        v = self.L.orc_sdf_swept(self.ctx, px, py, C.byref(t), _p(g))
        return v, t.value, g

    def true_sdf(self, px, py):
        t = C.c_double(0.0)
        g = np.zeros(3)
        v = self.L.orc_true_sdf(self.ctx, px, py, C.byref(t), _p(g))
        return v, t.value, g

    def query(self, xyz, nthreads=1):
        xyz = _f64(xyz).reshape(-1, 3)
        P = len(xyz)
        sdf, ts, g = np.zeros(P), np.zeros(P), np.zeros((P, 2))
        self.L.orc_query(self.ctx, _p(xyz), P, nthreads, _p(sdf), _p(ts), _p(g))
        return sdf, ts, g

    def penalty(self, xyz, nthreads=1, sum_mode=1, cost0=0.0, gradT0=None, gradC0=None, per_point=False):
        xyz = _f64(xyz).reshape(-1, 3)
        P, N = len(xyz), self.N
        cost = C.c_double(cost0)
        gT = np.zeros(N) if gradT0 is None else _f64(gradT0).copy()
        gC = np.zeros(18 * N) if gradC0 is None else np.asfortranarray(_f64(gradC0)).ravel(order="F").copy()
        sdf = ts = pc = None
        if per_point:
            sdf, ts, pc = np.zeros(P), np.zeros(P), np.zeros(P)
        self.L.orc_penalty(self.ctx, _p(xyz), P, nthreads, sum_mode, C.byref(cost), _p(gT), _p(gC),
                           _p(sdf), _p(ts), _p(pc))
        gC2 = gC.reshape(3, 6 * N).T.copy()  # (6N, 3)
        if per_point:
            return cost.value, gT, gC2, sdf, ts, pc
        return cost.value, gT, gC2

    # ---- SURVEY.md §8 row f3 ----
    def check_sub_sw_collision(self, father, child, pts_xy):
        f, c = _f64(father).reshape(3), _f64(child).reshape(3)
        q = _f64(pts_xy).reshape(-1, 2)
        return bool(self.L.orc_check_sub_sw_collision(self.ctx, _p(f), _p(c), _p(q), len(q)))

    def shape_kernels(self, kernel_size, kernel_count, resolution, safemargin):
        ks, K = int(kernel_size), int(kernel_count)
        m = np.zeros((K, ks, ks), dtype=np.uint8)
        b = np.zeros((K, ks, (ks + 7) // 8), dtype=np.uint8)
        yaws = np.zeros(K)
        u8 = C.POINTER(C.c_ubyte)
        n = self.L.orc_shape_kernels(self.ctx, ks, K, float(resolution), float(safemargin),
                                     m.ctypes.data_as(u8), b.ctypes.data_as(u8), _p(yaws))
        return m.astype(bool), b, yaws, n

    def set_trig_mode(self, mode):
        """1: sin/cos/atan2 by the ROCm device library's algorithms (diagnostic); 0: libm (oracle of record)."""
        self.L.orc_set_trig_mode(self.ctx, int(mode))

    def set_trig_perturb(self, seed):
        """Third oracle of the sensitivity bracket: libm sin/cos/atan2 moved by -1/0/+1 ulp (hash of argument and seed)."""
        self.L.orc_set_trig_perturb(self.ctx, int(seed))

    def set_modes(self, trig_mode, cum_locate):
        """The two diagnostic switches separately: trig_mode 1 = device-library sin/cos/atan2; cum_locate 1 = piece-local
        time as t - S_i (one rounding) instead of the reference's successive subtractions (TRJ:498-516)."""
        self.L.orc_set_modes(self.ctx, int(trig_mode), int(cum_locate))

    def counters(self):
        c = Counters()
        self.L.orc_get_counters(self.ctx, C.byref(c))
        return {k: getattr(c, k) for k, _ in Counters._fields_}

    def cost_function(self, xyz, x, nthreads=1):
        xyz = _f64(xyz).reshape(-1, 3)
        x = _f64(x)
        g = np.zeros_like(x)
        c3 = np.zeros(3)
        f = self.L.orc_cost_function(self.ctx, _p(xyz), len(xyz), nthreads, _p(x), _p(g), len(x), _p(c3))
        self.N = (len(x) + 3) // 4
        return f, g, c3


def shape_sdf(shape, x, y, poly_params=(0.0, 0.0, 0.0), polygon=None):
    """Raw shape SDF (getonlySDF(pos_rel)) through the oracle: a zero-motion trajectory."""
    o = _shape_oracle(shape, poly_params, polygon)
    return o.sdf_at_time(x, y, 0.0)


_SHAPE_CACHE = {}


def _shape_oracle(shape, poly_params, polygon):
    key = (shape, tuple(poly_params), None if polygon is None else tuple(np.ravel(polygon)))
    if key not in _SHAPE_CACHE:
        o = Oracle(shape, poly_params=poly_params, polygon=polygon)
        o.set_traj(np.zeros((6, 3)), np.array([1.0]))  # pose = identity for all t
        _SHAPE_CACHE[key] = o
    return _SHAPE_CACHE[key]


def minco_coeffs(head_state, tail_state, inPs, T):
    """(3x3 head, 3x3 tail, (N-1,3) waypoints, (N,) durations) -> (6N,3) coefficients."""
    L = lib()
    T = _f64(T)
    N = len(T)
    hs = np.asfortranarray(_f64(head_state)).ravel(order="F").copy()
    ts = np.asfortranarray(_f64(tail_state)).ravel(order="F").copy()
    q = _f64(inPs).reshape(-1, 3).copy()  # row i = waypoint i == column i of the 3x(N-1) matrix
    out = np.zeros(18 * N)
    L.orc_minco_coeffs(_p(hs), _p(ts), N, _p(q), _p(T), _p(out))
    return out.reshape(3, 6 * N).T.copy()


def forward_T(tau):
    tau = _f64(tau)
    T = np.zeros_like(tau)
    lib().orc_forward_T(_p(tau), _p(T), len(tau))
    return T


def backward_T(T):
    T = _f64(T)
    tau = np.zeros_like(T)
    lib().orc_backward_T(_p(T), _p(tau), len(T))
    return tau


def smoothed_l1(x, mu=0.01):
    f, df = C.c_double(0.0), C.c_double(0.0)
    ok = lib().orc_smoothed_l1(x, mu, C.byref(f), C.byref(df))
    return bool(ok), f.value, df.value


def map_points(cloud, centres, halfbd, resolution=1.0, sta_threshold=1):
    """Query-point producer (PCSmapManager + plan_manager waypoint loop): returns (points (n,3), dims)."""
    L = lib()
    cl = np.ascontiguousarray(cloud, dtype=np.float32).reshape(-1, 3)
    c = _f64(centres).reshape(-1, 3)
    hb = _f64(halfbd).reshape(3)
    dims = (C.c_int * 3)()
    cap = 1 << 22
    out = np.zeros((cap, 3))
    n = L.orc_map_points(cl.ctypes.data_as(C.POINTER(C.c_float)), len(cl), float(resolution), int(sta_threshold),
                         _p(c), len(c), _p(hb), _p(out), cap, dims)
    return out[:n].copy(), tuple(dims)
