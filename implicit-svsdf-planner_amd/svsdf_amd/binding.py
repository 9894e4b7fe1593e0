"""ctypes binding of libsvsdf_hip.so (C ABI declared in include/svsdf_c.h)."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

SHAPES = ["sdUnevenCapsule", "sdCutDisk", "sdTrapezoid", "sdRhombus", "star", "sdTunnel",
          "sdHorseshoe", "sdHeart", "sdOrientedVesica", "sdRoundedCross", "sdRoundedX", "bigX",
          "sdMoon", "sdPie", "sdPie2", "sdArc", "Polygon"]
SHAPE_ID = {n: i for i, n in enumerate(SHAPES)}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# every symbol include/svsdf_c.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "svsdf_create", "svsdf_destroy", "svsdf_config_default", "svsdf_shape_id_from_inputdata",
    "svsdf_shape_name", "svsdf_last_error_string", "svsdf_set_points", "svsdf_set_points_device",
    "svsdf_num_points", "svsdf_eval_penalty", "svsdf_eval_penalty_partial",
    "svsdf_accumulate_partial", "svsdf_lmbm_evaluate", "svsdf_last_costs", "svsdf_lmbm_begin",
    "svsdf_lmbm_finish", "svsdf_minco_coeffs", "svsdf_forward_T", "svsdf_backward_T",
    "svsdf_query_points", "svsdf_last_stats", "svsdf_shard_indices", "svsdf_set_profiling",
    "svsdf_shard_plan", "svsdf_lmbm_prepare", "svsdf_debug_sincos_mismatches",
    "svsdf_map_create", "svsdf_map_destroy", "svsdf_map_info", "svsdf_map_gather", "svsdf_pcd_read_ascii",
    "svsdf_check_sub_sw_collision", "svsdf_shape_kernels",
    "svsdf_lbfgs_params_default", "svsdf_lbfgs_minimize", "svsdf_optimize_traj",
    "svsdf_set_conditions", "svsdf_sum_partials", "svsdf_shape_bound",
    "svsdf_mesh_outline", "svsdf_mesh_outline_obj", "svsdf_swept_outline", "svsdf_outline_extrude",
    "svsdf_get_plan", "svsdf_set_plan", "svsdf_set_combine", "svsdf_group_info", "svsdf_debug_sdf_at",
    "svsdf_group_stripe", "svsdf_set_group_serial", "svsdf_shape_selfcheck", "svsdf_mesh_section", "svsdf_mesh_section_obj",
]


class LbfgsParams(C.Structure):
    """svsdf_lbfgs_params (include/svsdf_c.h), field for field lbfgs_parameter_t of the reference."""
    _fields_ = [("mem_size", C.c_int), ("g_epsilon", C.c_double), ("past", C.c_int), ("delta", C.c_double),
                ("max_iterations", C.c_int), ("max_linesearch", C.c_int), ("min_step", C.c_double),
                ("max_step", C.c_double), ("f_dec_coeff", C.c_double), ("s_curv_coeff", C.c_double),
                ("cautious_factor", C.c_double), ("machine_prec", C.c_double)]


EVALUATE_T = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)
PROGRESS_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                         C.c_int, C.c_int, C.c_int)
LBFGS_STATUS = {0: "CONVERGENCE", 1: "STOP", 2: "CANCELED", -1012: "INVALID_FUNCVAL", -1011: "MINIMUMSTEP",
                -1010: "MAXIMUMSTEP", -1009: "MAXIMUMLINESEARCH", -1008: "MAXIMUMITERATION", -1007: "WIDTHTOOSMALL",
                -1006: "INVALIDPARAMETERS", -1005: "INCREASEGRADIENT"}


class Config(C.Structure):
    _fields_ = [("shape_id", C.c_int), ("poly_params", C.c_double * 3), ("safety_hor", C.c_double),
                ("weight_p", C.c_double), ("rho", C.c_double), ("head_state", C.c_double * 9),
                ("tail_state", C.c_double * 9), ("device", C.c_int), ("polygon_nverts", C.c_int),
                ("polygon_xy", _dp), ("rank", C.c_int), ("world_size", C.c_int), ("flags", C.c_int),
                ("n_devices", C.c_int), ("devices", C.c_int * 8), ("combine", C.c_int),
                ("polygon_nloops", C.c_int), ("polygon_loop_sizes", C.POINTER(C.c_int))]


class Stats(C.Structure):
    _fields_ = [("points", C.c_ulonglong), ("interior_points", C.c_ulonglong),
                ("solves", C.c_ulonglong), ("gsip_samples", C.c_ulonglong), ("sdf_evals", C.c_ulonglong),
                ("scan_evals", C.c_ulonglong), ("device_ms", C.c_double), ("solve_ms", C.c_double),
                ("solve_launches", C.c_uint), ("gsip_iterations", C.c_uint), ("culled_points", C.c_ulonglong),
                ("gsip_bound_mode", C.c_int), ("bound_mode_decided", C.c_int), ("bound_ratio", C.c_double),
                ("n_devices", C.c_int), ("combine", C.c_int), ("combine_ms", C.c_double), ("setup_ms", C.c_double),
                ("piece_time_exact", C.c_int), ("solve_ms_sum", C.c_double), ("round_scan_evals", C.c_ulonglong),
                ("round_ms", C.c_double), ("round_ms_sum", C.c_double), ("batches", C.c_int),
                ("speculative_evals", C.c_ulonglong), ("plan_settled", C.c_int), ("tail_iter", C.c_int),
                ("tail_launches", C.c_uint), ("tail_points", C.c_ulonglong), ("tail_ms", C.c_double),
                ("tail_ms_sum", C.c_double), ("shader_clock_mhz", C.c_double), ("fanout_ms", C.c_double)]


class Plan(C.Structure):
    _fields_ = [("bound_mode", C.c_int), ("batches", C.c_int), ("lanes_per_query", C.c_int), ("tail_iter", C.c_int),
                ("settled", C.c_int)]


PLAN_AUTO = -1


class OutlineStats(C.Structure):
    _fields_ = [("nodes_evaluated", C.c_ulonglong), ("dense_nodes", C.c_ulonglong), ("cells_marched", C.c_ulonglong),
                ("batches", C.c_ulonglong), ("open_chains", C.c_int)]


class SvsdfError(RuntimeError):
    pass


def lib_path():
    """In-tree HIP library.  SVSDF_LIB_VARIANT=<v> loads libsvsdf_hip_<v>.so instead (experimental builds
    made by tools/ scripts for A/B measurements; nothing in tests/ or bench.py sets it)."""
    v = os.environ.get("SVSDF_LIB_VARIANT", "")
    return os.path.join(_PKG, "libsvsdf_hip_%s.so" % v if v else "libsvsdf_hip.so")


def lib():
    """Load the HIP library.  Raises if it has not been built: there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise SvsdfError(f"{path} is missing: run `python __graft_entry__.py build` "
                         "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    L = C.CDLL(path)
    L.svsdf_create.restype = C.c_void_p
    L.svsdf_create.argtypes = [C.POINTER(Config)]
    L.svsdf_destroy.argtypes = [C.c_void_p]
    L.svsdf_config_default.argtypes = [C.POINTER(Config)]
    L.svsdf_shape_id_from_inputdata.restype = C.c_int
    L.svsdf_shape_id_from_inputdata.argtypes = [C.c_char_p]
    L.svsdf_shape_name.restype = C.c_char_p
    L.svsdf_shape_name.argtypes = [C.c_int]
    L.svsdf_last_error_string.restype = C.c_char_p
    L.svsdf_last_error_string.argtypes = [C.c_void_p]
    L.svsdf_set_points.argtypes = [C.c_void_p, _dp, C.c_size_t]
    L.svsdf_set_points_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.svsdf_num_points.restype = C.c_size_t
    L.svsdf_num_points.argtypes = [C.c_void_p]
    L.svsdf_eval_penalty.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
    L.svsdf_eval_penalty_partial.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_size_t)]
    L.svsdf_accumulate_partial.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
    L.svsdf_sum_partials.argtypes = [_dp, C.c_int, C.c_size_t, _dp]
    L.svsdf_set_conditions.argtypes = [C.c_void_p, _dp, _dp]
    L.svsdf_shape_bound.argtypes = [C.c_void_p, _dp]
    L.svsdf_lmbm_evaluate.restype = C.c_double
    L.svsdf_lmbm_evaluate.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
    L.svsdf_last_costs.argtypes = [C.c_void_p, _dp]
    L.svsdf_lmbm_begin.argtypes = [C.c_void_p, _dp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.svsdf_lmbm_finish.restype = C.c_double
    L.svsdf_lmbm_finish.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
    L.svsdf_minco_coeffs.argtypes = [_dp, _dp, C.c_int, _dp, _dp, _dp]
    L.svsdf_forward_T.argtypes = [_dp, _dp, C.c_int]
    L.svsdf_backward_T.argtypes = [_dp, _dp, C.c_int]
    L.svsdf_query_points.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
    L.svsdf_last_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.svsdf_shard_indices.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.svsdf_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.svsdf_shard_plan.argtypes = [_dp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong),
                                   C.POINTER(C.c_size_t)]
    L.svsdf_lmbm_prepare.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp]
    _fp = C.POINTER(C.c_float)
    L.svsdf_map_create.restype = C.c_void_p
    L.svsdf_map_create.argtypes = [_fp, C.c_size_t, C.c_double, C.c_int]
    L.svsdf_map_destroy.argtypes = [C.c_void_p]
    L.svsdf_map_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), _dp, _dp, C.POINTER(C.c_size_t)]
    L.svsdf_map_gather.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.svsdf_pcd_read_ascii.argtypes = [C.c_char_p, _fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.svsdf_debug_sincos_mismatches.restype = C.c_longlong
    L.svsdf_debug_sincos_mismatches.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
    L.svsdf_lbfgs_params_default.argtypes = [C.POINTER(LbfgsParams)]
    L.svsdf_lbfgs_params_default.restype = None
    _ip = C.POINTER(C.c_int)
    L.svsdf_lbfgs_minimize.argtypes = [C.c_int, _dp, EVALUATE_T, C.c_void_p, PROGRESS_T, C.c_void_p,
                                       C.POINTER(LbfgsParams), _dp, _ip, _ip]
    L.svsdf_optimize_traj.argtypes = [C.c_void_p, _dp, C.c_int, C.POINTER(LbfgsParams), PROGRESS_T, C.c_void_p,
                                      _dp, _ip, _ip]
    _u8p = C.POINTER(C.c_ubyte)
    L.svsdf_check_sub_sw_collision.argtypes = [C.c_void_p, C.c_size_t, _dp, _dp, C.POINTER(C.c_size_t), _dp, _u8p]
    L.svsdf_shape_kernels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, _u8p, _u8p, _dp,
                                      C.POINTER(C.c_int)]
    L.svsdf_mesh_outline.argtypes = [_dp, C.c_size_t, _ip, C.c_size_t, C.c_double, _dp, C.c_size_t,
                                     C.POINTER(C.c_size_t), _ip]
    L.svsdf_mesh_outline_obj.argtypes = [C.c_char_p, C.c_double, _dp, C.c_size_t, C.POINTER(C.c_size_t), _ip]
    L.svsdf_swept_outline.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_double, C.c_double, _dp, C.c_size_t,
                                      C.POINTER(C.c_size_t), _ip, C.c_size_t, C.POINTER(C.c_size_t),
                                      C.POINTER(OutlineStats)]
    if hasattr(L, "svsdf_get_plan"):   # (absent only from A/B libraries built from older commits, tools/exp_variants.py)
        L.svsdf_get_plan.argtypes = [C.c_void_p, C.POINTER(Plan)]
        L.svsdf_set_plan.argtypes = [C.c_void_p, C.POINTER(Plan)]
        L.svsdf_set_combine.argtypes = [C.c_void_p, C.c_int]
        L.svsdf_group_info.argtypes = [C.c_void_p, _ip, _ip, _ip]
    if hasattr(L, "svsdf_group_stripe"):
        L.svsdf_group_stripe.argtypes = [C.c_void_p, C.c_int, _ip, C.POINTER(C.c_size_t), C.POINTER(Stats), C.POINTER(Plan)]
        L.svsdf_set_group_serial.argtypes = [C.c_void_p, C.c_int]
    _LIB = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _colmajor(m):
    """(rows, cols) array -> flat column-major copy (Eigen's default storage)."""
    return np.asfortranarray(_f64(m)).ravel(order="F").copy()


def shape_id_from_inputdata(inputdata):
    return lib().svsdf_shape_id_from_inputdata(inputdata.encode())


def minco_coeffs(head_state, tail_state, inPs, T):
    """(3x3 head, 3x3 tail, (N-1, 3) waypoints, (N,) durations) -> (6N, 3) coefficients
    (row 6i+k = coefficient of s^k of piece i), via the product's host MINCO."""
    T = _f64(T)
    N = len(T)
    q = _f64(inPs).reshape(-1, 3).copy()
    out = np.zeros(18 * N)
    rc = lib().svsdf_minco_coeffs(_p(_colmajor(head_state)), _p(_colmajor(tail_state)), N, _p(q), _p(T), _p(out))
    if rc:
        raise SvsdfError(f"svsdf_minco_coeffs failed: {rc}")
    return out.reshape(3, 6 * N).T.copy()


def mesh_outline(V, F, z0=0.0):
    """z = z0 cross-section of a triangle mesh (V (nv, 3), F (nf, 3) zero-based) -> (outline (n, 2), closed loops
    found); pure host (svsdf_mesh_outline).  The outline is what `polygon=` of SvsdfContext takes: BASELINE config 5."""
    V = _f64(V).reshape(-1, 3)
    F = np.ascontiguousarray(F, dtype=np.int32).reshape(-1, 3)
    n, loops = C.c_size_t(), C.c_int()
    ip = C.POINTER(C.c_int)
    rc = lib().svsdf_mesh_outline(_p(V), len(V), F.ctypes.data_as(ip), len(F), float(z0), None, 0, C.byref(n), C.byref(loops))
    if rc:
        raise SvsdfError(f"svsdf_mesh_outline failed: {rc}")
    xy = np.zeros((n.value, 2))
    lib().svsdf_mesh_outline(_p(V), len(V), F.ctypes.data_as(ip), len(F), float(z0), _p(xy), n.value, C.byref(n), C.byref(loops))
    return xy, loops.value


def mesh_section(V, F, z0=0.0):
    """All closed loops of the z = z0 section of a triangle mesh (svsdf_mesh_section): (xy (n, 2) loop after loop, largest
    enclosed area first; loop sizes)."""
    L = lib()
    V = _f64(V).reshape(-1, 3)
    F = np.ascontiguousarray(F, dtype=np.int32).reshape(-1, 3)
    n, nl = C.c_size_t(), C.c_size_t()
    ip = C.POINTER(C.c_int)
    L.svsdf_mesh_section.argtypes = [_dp, C.c_size_t, ip, C.c_size_t, C.c_double, _dp, C.c_size_t, C.POINTER(C.c_size_t), ip,
                                     C.c_size_t, C.POINTER(C.c_size_t)]
    rc = L.svsdf_mesh_section(_p(V), len(V), F.ctypes.data_as(ip), len(F), float(z0), None, 0, C.byref(n), None, 0, C.byref(nl))
    if rc:
        raise SvsdfError("svsdf_mesh_section failed: " + L.svsdf_last_error_string(None).decode())
    xy = np.zeros((n.value, 2))
    sizes = np.zeros(nl.value, dtype=np.int32)
    rc = L.svsdf_mesh_section(_p(V), len(V), F.ctypes.data_as(ip), len(F), float(z0), _p(xy), n.value, C.byref(n),
                              sizes.ctypes.data_as(ip), nl.value, C.byref(nl))
    if rc:
        raise SvsdfError("svsdf_mesh_section failed: " + L.svsdf_last_error_string(None).decode())
    return xy, [int(v) for v in sizes]


def mesh_section_obj(path, z0=0.0):
    """The same for a Wavefront .obj file (svsdf_mesh_section_obj)."""
    L = lib()
    n, nl = C.c_size_t(), C.c_size_t()
    ip = C.POINTER(C.c_int)
    L.svsdf_mesh_section_obj.argtypes = [C.c_char_p, C.c_double, _dp, C.c_size_t, C.POINTER(C.c_size_t), ip, C.c_size_t,
                                         C.POINTER(C.c_size_t)]
    path = os.fspath(path)
    rc = L.svsdf_mesh_section_obj(path.encode(), float(z0), None, 0, C.byref(n), None, 0, C.byref(nl))
    if rc:
        raise SvsdfError("svsdf_mesh_section_obj failed: " + L.svsdf_last_error_string(None).decode())
    xy = np.zeros((n.value, 2))
    sizes = np.zeros(nl.value, dtype=np.int32)
    rc = L.svsdf_mesh_section_obj(path.encode(), float(z0), _p(xy), n.value, C.byref(n), sizes.ctypes.data_as(ip), nl.value,
                                  C.byref(nl))
    if rc:
        raise SvsdfError("svsdf_mesh_section_obj failed: " + L.svsdf_last_error_string(None).decode())
    return xy, [int(v) for v in sizes]


def mesh_outline_obj(path, z0=0.0):
    """Same from a Wavefront .obj file (svsdf_mesh_outline_obj)."""
    n, loops = C.c_size_t(), C.c_int()
    rc = lib().svsdf_mesh_outline_obj(str(path).encode(), float(z0), None, 0, C.byref(n), C.byref(loops))
    if rc:
        raise SvsdfError(f"svsdf_mesh_outline_obj({path}) failed: {rc}")
    xy = np.zeros((n.value, 2))
    lib().svsdf_mesh_outline_obj(str(path).encode(), float(z0), _p(xy), n.value, C.byref(n), C.byref(loops))
    return xy, loops.value


def outline_extrude(loops, z0=-0.5, z1=0.5, caps=True):
    """Surface of the extrusion of closed polylines (svsdf_outline_extrude): V (nv, 3), F (nf, 3); walls and, with caps,
    the bottom and top faces (a closed surface).  Host only."""
    xy = _f64(np.vstack(loops))
    sizes = np.ascontiguousarray([len(lp) for lp in loops], dtype=np.int32)
    nv, nf = C.c_size_t(), C.c_size_t()
    L = lib()
    L.svsdf_outline_extrude.argtypes = [_dp, _ip, C.c_size_t, C.c_double, C.c_double, C.c_int, _dp, C.c_size_t,
                                        C.POINTER(C.c_size_t), _ip, C.c_size_t, C.POINTER(C.c_size_t)]
    rc = L.svsdf_outline_extrude(_p(xy), sizes.ctypes.data_as(_ip), len(sizes), z0, z1, int(bool(caps)), None, 0,
                                 C.byref(nv), None, 0, C.byref(nf))
    if rc:
        raise SvsdfError(f"svsdf_outline_extrude failed: {rc}")
    V = np.zeros((nv.value, 3))
    F = np.zeros((nf.value, 3), dtype=np.int32)
    L.svsdf_outline_extrude(_p(xy), sizes.ctypes.data_as(_ip), len(sizes), z0, z1, int(bool(caps)), _p(V), nv.value,
                            C.byref(nv), F.ctypes.data_as(_ip), nf.value, C.byref(nf))
    return V, F


def forward_T(tau):
    tau = _f64(tau)
    T = np.zeros_like(tau)
    lib().svsdf_forward_T(_p(tau), _p(T), len(tau))
    return T


def backward_T(T):
    T = _f64(T)
    tau = np.zeros_like(T)
    lib().svsdf_backward_T(_p(T), _p(tau), len(T))
    return tau


class OccupancyMap:
    """Host-side query-point producer (PCSmapManager + the plan_manager waypoint loop)."""

    def __init__(self, cloud_xyz, resolution=1.0, sta_threshold=1):
        self.L = lib()
        pts = np.ascontiguousarray(cloud_xyz, dtype=np.float32).reshape(-1, 3)
        h = self.L.svsdf_map_create(pts.ctypes.data_as(C.POINTER(C.c_float)), len(pts), float(resolution), int(sta_threshold))
        if not h:
            raise SvsdfError("svsdf_map_create failed")
        self.h = C.c_void_p(h)

    @staticmethod
    def read_pcd(path):
        L = lib()
        n = C.c_size_t()
        if L.svsdf_pcd_read_ascii(path.encode(), None, 0, C.byref(n)):
            raise SvsdfError(f"cannot read {path} as ASCII PCD (FIELDS x y z)")
        xyz = np.zeros((n.value, 3), dtype=np.float32)
        L.svsdf_pcd_read_ascii(path.encode(), xyz.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n))
        return xyz

    def info(self):
        dims = (C.c_int * 3)()
        bmin, bmax = np.zeros(3), np.zeros(3)
        occ = C.c_size_t()
        self.L.svsdf_map_info(self.h, dims, _p(bmin), _p(bmax), C.byref(occ))
        return dict(dims=tuple(dims), bmin=bmin, bmax=bmax, occupied=occ.value)

    def gather(self, centres, halfbd):
        c = _f64(centres).reshape(-1, 3)
        hb = _f64(halfbd).reshape(3)
        n = C.c_size_t()
        rc = self.L.svsdf_map_gather(self.h, _p(c), len(c), _p(hb), None, 0, C.byref(n))
        if rc:
            raise SvsdfError(f"svsdf_map_gather failed: {rc}")
        out = np.zeros((n.value, 3))
        if n.value:
            self.L.svsdf_map_gather(self.h, _p(c), len(c), _p(hb), _p(out), n.value, C.byref(n))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.svsdf_map_destroy(self.h)
                self.h = None
        except Exception:
            pass


FLAG_KEEP_INPUT_ORDER = 1
FLAG_HOST_ONLY = 2
FLAG_EXACT_PIECE_TIME = 4
FLAG_FAST_PIECE_TIME = 8
COMBINE_AUTO, COMBINE_HOST, COMBINE_RCCL = 0, 1, 2


def sum_partials(partials):
    """(G, len) per-device partials -> their fixed-order sum (the multi-device context's host combine)."""
    p = np.ascontiguousarray(partials, dtype=np.float64)
    out = np.zeros(p.shape[1])
    rc = lib().svsdf_sum_partials(_p(p), p.shape[0], p.shape[1], _p(out))
    if rc:
        raise SvsdfError(f"svsdf_sum_partials failed: {rc}")
    return out


def shard_plan(xyz, rank, world_size, flags=0):
    """Original indices owned by `rank` (pure host; same plan svsdf_set_points uses)."""
    xyz = _f64(xyz).reshape(-1, 3)
    n = C.c_size_t()
    idx = np.zeros(len(xyz), dtype=np.int64)
    rc = lib().svsdf_shard_plan(_p(xyz), len(xyz), rank, world_size, flags,
                                idx.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(n))
    if rc:
        raise SvsdfError(f"svsdf_shard_plan failed: {rc}")
    return idx[:n.value].copy()


def lbfgs_params(**kw):
    p = LbfgsParams()
    lib().svsdf_lbfgs_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown L-BFGS parameter %r" % k)
        setattr(p, k, v)
    return p


def _wrap_progress(progress):
    if progress is None:
        return C.cast(None, PROGRESS_T)

    def _cb(_user, x, g, fx, step, n, k, ls):
        xs = np.ctypeslib.as_array(x, shape=(n,)).copy()
        gs = np.ctypeslib.as_array(g, shape=(n,)).copy()
        return int(bool(progress(xs, gs, fx, step, k, ls)))
    return PROGRESS_T(_cb)


def lbfgs_minimize(fun, x0, progress=None, **params):
    """Generic driver over a Python callable fun(x) -> (f, g).  Returns (x, f, status, iterations, evaluations)."""
    x = _f64(x0).copy()
    n = len(x)

    def _ev(_inst, xp, gp, nn):
        f, g = fun(np.ctypeslib.as_array(xp, shape=(nn,)).copy())
        np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
        return float(f)
    p = lbfgs_params(**params)
    f, it, ev = C.c_double(0.0), C.c_int(0), C.c_int(0)
    rc = lib().svsdf_lbfgs_minimize(n, _p(x), EVALUATE_T(_ev), None, _wrap_progress(progress), None, C.byref(p),
                                    C.byref(f), C.byref(it), C.byref(ev))
    return x, f.value, rc, it.value, ev.value


class SvsdfContext:
    """Owns one svsdf_ctx (one GPU, one shard of the query points)."""

    def __init__(self, shape="star", safety_hor=0.7, weight_p=60.0, rho=3.8,
                 poly_params=(0.0, 0.0, 0.0), polygon=None, head_state=None, tail_state=None,
                 device=-1, rank=0, world_size=1, flags=0, devices=None, combine=COMBINE_AUTO, polygon_loops=None):
        """polygon_loops: vertex counts of the closed loops `polygon` is made of (mesh_section); None: one loop."""
        self.L = lib()
        cfg = Config()
        self.L.svsdf_config_default(C.byref(cfg))
        cfg.shape_id = SHAPE_ID[shape] if isinstance(shape, str) else int(shape)
        cfg.poly_params[:] = list(map(float, poly_params))
        cfg.safety_hor, cfg.weight_p, cfg.rho = float(safety_hor), float(weight_p), float(rho)
        hs = np.zeros((3, 3)) if head_state is None else _f64(head_state)
        ts = np.zeros((3, 3)) if tail_state is None else _f64(tail_state)
        cfg.head_state[:] = list(_colmajor(hs))
        cfg.tail_state[:] = list(_colmajor(ts))
        cfg.device, cfg.rank, cfg.world_size, cfg.flags = int(device), int(rank), int(world_size), int(flags)
        if devices is not None and (len(devices) >= 2 or int(combine) == COMBINE_RCCL):    # in-process multi-GPU
            cfg.n_devices = len(devices)
            for k, d in enumerate(devices):
                cfg.devices[k] = int(d)
        elif devices is not None and len(devices) == 1:                                    # a list of one: that device
            cfg.device = int(devices[0])
        cfg.combine = int(combine)
        self._poly = None
        if polygon is not None:
            self._poly = _f64(polygon).reshape(-1, 2).copy()
            cfg.polygon_nverts = len(self._poly)
            cfg.polygon_xy = _p(self._poly)
            if polygon_loops is not None and len(polygon_loops) >= 2:
                self._loops = (C.c_int * len(polygon_loops))(*[int(v) for v in polygon_loops])
                cfg.polygon_nloops = len(polygon_loops)
                cfg.polygon_loop_sizes = C.cast(self._loops, C.POINTER(C.c_int))
        h = self.L.svsdf_create(C.byref(cfg))
        if not h:
            raise SvsdfError("svsdf_create failed: " + self.L.svsdf_last_error_string(None).decode())
        self.ctx = C.c_void_p(h)
        self.head_state, self.tail_state = hs, ts

    def close(self):
        if getattr(self, "ctx", None):
            self.L.svsdf_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise SvsdfError(f"{what} failed ({rc}): " + self.L.svsdf_last_error_string(self.ctx).decode())

    def set_conditions(self, head_state, tail_state):
        hs, ts = _f64(head_state).reshape(3, 3), _f64(tail_state).reshape(3, 3)
        self._chk(self.L.svsdf_set_conditions(self.ctx, _p(_colmajor(hs)), _p(_colmajor(ts))), "svsdf_set_conditions")
        self.head_state, self.tail_state = hs, ts

    # ---- points ----
    def set_points(self, xyz):
        xyz = _f64(xyz).reshape(-1, 3)
        self._chk(self.L.svsdf_set_points(self.ctx, _p(xyz), len(xyz)), "svsdf_set_points")

    def set_points_device(self, dev_ptr, P):
        self._chk(self.L.svsdf_set_points_device(self.ctx, C.c_void_p(int(dev_ptr)), int(P)), "svsdf_set_points_device")

    def num_points(self):
        return int(self.L.svsdf_num_points(self.ctx))

    def shard_indices(self):
        idx = np.zeros(self.num_points(), dtype=np.int64)
        if len(idx):
            self._chk(self.L.svsdf_shard_indices(self.ctx, idx.ctypes.data_as(C.POINTER(C.c_longlong))), "svsdf_shard_indices")
        return idx

    # ---- inner operator ----
    def eval_penalty(self, coeffs, T, cost0=0.0, gradT0=None, gradC0=None):
        """coeffs (6N, 3), T (N,) -> (cost, gradT (N,), gradC (6N, 3)); accumulates like BEO:774-869."""
        T = _f64(T)
        N = len(T)
        cm = _colmajor(coeffs)
        cost = C.c_double(cost0)
        gT = np.zeros(N) if gradT0 is None else _f64(gradT0).copy()
        gC = np.zeros(18 * N) if gradC0 is None else _colmajor(gradC0)
        self._chk(self.L.svsdf_eval_penalty(self.ctx, N, _p(cm), _p(T), C.byref(cost), _p(gT), _p(gC)), "svsdf_eval_penalty")
        return cost.value, gT, gC.reshape(3, 6 * N).T.copy()

    def eval_penalty_partial(self, coeffs, T):
        """Runs the device pipeline; returns (device pointer, length) of [cost, gradC, gradT]."""
        T = _f64(T)
        N = len(T)
        cm = _colmajor(coeffs)
        ptr, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.svsdf_eval_penalty_partial(self.ctx, N, _p(cm), _p(T), C.byref(ptr), C.byref(n)), "svsdf_eval_penalty_partial")
        return ptr.value, n.value

    def accumulate_partial(self, N, partial, cost0=0.0, gradT0=None, gradC0=None):
        partial = _f64(partial)
        cost = C.c_double(cost0)
        gT = np.zeros(N) if gradT0 is None else _f64(gradT0).copy()
        gC = np.zeros(18 * N) if gradC0 is None else _colmajor(gradC0)
        self._chk(self.L.svsdf_accumulate_partial(self.ctx, N, _p(partial), C.byref(cost), _p(gT), _p(gC)), "svsdf_accumulate_partial")
        return cost.value, gT, gC.reshape(3, 6 * N).T.copy()

    # ---- full callback ----
    def lmbm_evaluate(self, x):
        x = _f64(x)
        g = np.zeros_like(x)
        f = self.L.svsdf_lmbm_evaluate(self.ctx, _p(x), _p(g), len(x))
        if not np.isfinite(f):
            raise SvsdfError("svsdf_lmbm_evaluate failed: " + self.L.svsdf_last_error_string(self.ctx).decode())
        return f, g

    def last_costs(self):
        c = np.zeros(3)
        self.L.svsdf_last_costs(self.ctx, _p(c))
        return c

    def lmbm_begin(self, x):
        x = _f64(x)
        ptr, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.svsdf_lmbm_begin(self.ctx, _p(x), len(x), C.byref(ptr), C.byref(n)), "svsdf_lmbm_begin")
        return ptr.value, n.value

    def lmbm_prepare(self, x):
        """Host half of lmbm_begin: returns (coeffs (6N, 3), T (N,)) the device stage receives."""
        x = _f64(x)
        N = (len(x) + 3) // 4
        cm, T = np.zeros(18 * N), np.zeros(N)
        self._chk(self.L.svsdf_lmbm_prepare(self.ctx, _p(x), len(x), _p(cm), _p(T)), "svsdf_lmbm_prepare")
        return cm.reshape(3, 6 * N).T.copy(), T

    def lmbm_finish(self, partial, n):
        partial = _f64(partial)
        g = np.zeros(n)
        f = self.L.svsdf_lmbm_finish(self.ctx, _p(partial), _p(g), n)
        if not np.isfinite(f):
            raise SvsdfError("svsdf_lmbm_finish failed: " + self.L.svsdf_last_error_string(self.ctx).decode())
        return f, g

    # ---- diagnostics ----
    def swept_outline(self, coeffs, T, cell=0.05, margin=0.0):
        """Boundary of the swept volume's z = 0 section (svsdf_swept_outline; what the reference's sw_calculate /
        calculateSwept are for): list of (n_i, 2) closed polylines, inside on the left, + the work statistics."""
        T = _f64(T)
        N = len(T)
        cm = _colmajor(coeffs)
        nv, nl, st = C.c_size_t(), C.c_size_t(), OutlineStats()
        self._chk(self.L.svsdf_swept_outline(self.ctx, N, _p(cm), _p(T), float(cell), float(margin), None, 0,
                                             C.byref(nv), None, 0, C.byref(nl), C.byref(st)), "svsdf_swept_outline")
        xy = np.zeros((nv.value, 2))
        sizes = np.zeros(max(nl.value, 1), dtype=np.int32)
        if nv.value:
            self._chk(self.L.svsdf_swept_outline(self.ctx, N, _p(cm), _p(T), float(cell), float(margin), _p(xy), nv.value,
                                                 C.byref(nv), sizes.ctypes.data_as(_ip), nl.value, C.byref(nl),
                                                 C.byref(st)), "svsdf_swept_outline")
        loops, off = [], 0
        for n in sizes[:nl.value]:
            loops.append(xy[off:off + int(n)].copy())
            off += int(n)
        return loops, {k: getattr(st, k) for k, _ in OutlineStats._fields_}

    def query_points(self, coeffs, T):
        """Per-point (sdf, t*, grad_xy) in the ORIGINAL order of the points given to set_points
        (world_size == 1) or of this rank's shard (see shard_indices)."""
        T = _f64(T)
        N = len(T)
        cm = _colmajor(coeffs)
        P = self.num_points()
        sdf, ts, g = np.zeros(P), np.zeros(P), np.zeros((P, 2))
        self._chk(self.L.svsdf_query_points(self.ctx, N, _p(cm), _p(T), _p(sdf), _p(ts), _p(g)), "svsdf_query_points")
        idx = self.shard_indices()
        order = np.argsort(idx, kind="stable")
        return sdf[order], ts[order], g[order], idx[order]

    # ---- optimizer driver ----
    def optimize_traj(self, x, progress=None, **params):
        """optimize_traj_lmbm analogue (back_end_optimizer.cpp:3-95) with the in-library L-BFGS driver.
        Returns (x_opt, final_cost, status, iterations, evaluations)."""
        x = _f64(x).copy()
        p = lbfgs_params(**params)
        f, it, ev = C.c_double(0.0), C.c_int(0), C.c_int(0)
        cb = _wrap_progress(progress)
        rc = self.L.svsdf_optimize_traj(self.ctx, _p(x), len(x), C.byref(p), cb, None, C.byref(f), C.byref(it),
                                        C.byref(ev))
        return x, f.value, rc, it.value, ev.value

    # ---- front end ----
    def check_sub_sw_collision(self, father_states, child_states, points_per_edge):
        """Batched SweptVolumeManager::checkSubSWCollision (sw_manager.hpp:1171-1211).

        father_states / child_states: (E, 3) arrays of (x, y, yaw); points_per_edge: sequence of E
        (n_e, 2) obstacle-point arrays.  Returns a bool array, True where the reference returns true."""
        fs = _f64(father_states).reshape(-1, 3)
        cs = _f64(child_states).reshape(-1, 3)
        E = len(fs)
        if len(cs) != E or len(points_per_edge) != E:
            raise ValueError("check_sub_sw_collision: one child state and one point set per edge")
        pts = [_f64(q).reshape(-1, 2) for q in points_per_edge]
        offs = np.zeros(E + 1, dtype=np.uintp)
        if E:
            offs[1:] = np.cumsum([len(q) for q in pts])
        flat = np.ascontiguousarray(np.concatenate(pts, axis=0)) if E and offs[-1] else np.zeros((0, 2))
        out = np.zeros(max(E, 1), dtype=np.uint8)
        self._chk(self.L.svsdf_check_sub_sw_collision(
            self.ctx, E, _p(fs), _p(cs), offs.ctypes.data_as(C.POINTER(C.c_size_t)), _p(flat),
            out.ctypes.data_as(C.POINTER(C.c_ubyte))), "svsdf_check_sub_sw_collision")
        return out[:E].astype(bool)

    def shape_kernels(self, kernel_size, kernel_count, resolution, safemargin):
        """BasicShape::initShape (Shape.hpp:386-430): returns (map (K, ks, ks) bool, bytes (K, ks, (ks+7)//8)
        uint8, yaws (K,), loop_count)."""
        ks, K = int(kernel_size), int(kernel_count)
        m = np.zeros((K, ks, ks), dtype=np.uint8)
        b = np.zeros((K, ks, (ks + 7) // 8), dtype=np.uint8)
        yaws = np.zeros(K)
        n = C.c_int(0)
        u8 = C.POINTER(C.c_ubyte)
        self._chk(self.L.svsdf_shape_kernels(self.ctx, ks, K, float(resolution), float(safemargin),
                                             m.ctypes.data_as(u8), b.ctypes.data_as(u8), _p(yaws), C.byref(n)),
                  "svsdf_shape_kernels")
        return m.astype(bool), b, yaws, n.value

    def debug_sdf_at(self, coeffs, T, points_xy, t):
        """SDF-at-time of (point, time) pairs on the device with its intermediates (svsdf_debug_sdf_at): (n, 8) array
        sdf, pose x, y, cos, sin, body-frame x, y, piece-time mode."""
        T = _f64(T)
        cm = _colmajor(coeffs)
        xy = _f64(points_xy).reshape(-1, 2).copy()
        tt = _f64(t).ravel().copy()
        out = np.zeros((len(tt), 8))
        self.L.svsdf_debug_sdf_at.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_size_t, _dp, _dp, _dp]
        self._chk(self.L.svsdf_debug_sdf_at(self.ctx, len(T), _p(cm), _p(T), len(tt), _p(xy), _p(tt), _p(out)), "svsdf_debug_sdf_at")
        return out

    def sincos_mismatches(self, lo, hi, n):
        return int(self.L.svsdf_debug_sincos_mismatches(self.ctx, float(lo), float(hi), int(n)))

    def shape_bound(self):
        """(analytic R with sdf(q) >= |q| - R, largest |q| - sdf(q) sampled at creation)."""
        o = np.zeros(2)
        self._chk(self.L.svsdf_shape_bound(self.ctx, _p(o)), "svsdf_shape_bound")
        return float(o[0]), float(o[1])

    def shape_selfcheck(self):
        """(R analytic, R sampled, Lipschitz excess: 0 = the shape SDF passed the 1-Lipschitz self-check)."""
        o = np.zeros(3)
        self.L.svsdf_shape_selfcheck.argtypes = [C.c_void_p, _dp]
        self._chk(self.L.svsdf_shape_selfcheck(self.ctx, _p(o)), "svsdf_shape_selfcheck")
        return float(o[0]), float(o[1]), float(o[2])

    def get_plan(self):
        """Launch plan in force (svsdf_get_plan): bound_mode, batches, lanes_per_query, tail_iter, settled."""
        pl = Plan()
        self._chk(self.L.svsdf_get_plan(self.ctx, C.byref(pl)), "svsdf_get_plan")
        return {k: getattr(pl, k) for k, _ in Plan._fields_}

    def set_plan(self, bound_mode=PLAN_AUTO, batches=PLAN_AUTO, lanes_per_query=PLAN_AUTO, tail_iter=PLAN_AUTO):
        """Pin fields of the launch plan (svsdf_set_plan; PLAN_AUTO leaves a field to its rule; batches=-2: measured).
        Only moves time: every plan returns the same bits."""
        pl = Plan(int(bound_mode), int(batches), int(lanes_per_query), int(tail_iter), 0)
        self._chk(self.L.svsdf_set_plan(self.ctx, C.byref(pl)), "svsdf_set_plan")

    def set_combine(self, combine):
        """Multi-device contexts: 'host' or 'rccl' sum of the devices' partials (svsdf_set_combine)."""
        code = {"host": 1, "rccl": 2}.get(combine, combine)
        self._chk(self.L.svsdf_set_combine(self.ctx, int(code)), "svsdf_set_combine")

    def group_info(self):
        """(devices, combine mode, ranks of the RCCL communicator as the communicator reports them; 0: none)."""
        nd, cb, rk = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.svsdf_group_info(self.ctx, C.byref(nd), C.byref(cb), C.byref(rk)), "svsdf_group_info")
        return {"n_devices": nd.value, "combine": ["auto", "host", "rccl"][cb.value], "rccl_ranks": rk.value}

    def group_stripe(self, k):
        """Stripe k of a multi-device context (svsdf_group_stripe): device, points, its stats and plan."""
        dev, pts, st, pl = C.c_int(), C.c_size_t(), Stats(), Plan()
        self._chk(self.L.svsdf_group_stripe(self.ctx, int(k), C.byref(dev), C.byref(pts), C.byref(st), C.byref(pl)), "svsdf_group_stripe")
        return {"device": dev.value, "points": pts.value, "stats": {n: getattr(st, n) for n, _ in Stats._fields_},
                "plan": {n: getattr(pl, n) for n, _ in Plan._fields_}}

    def set_group_serial(self, serial=True):
        """Diagnostic: the stripes of a multi-device context one after the other (svsdf_set_group_serial)."""
        self._chk(self.L.svsdf_set_group_serial(self.ctx, int(bool(serial))), "svsdf_set_group_serial")

    def set_profiling(self, enable=True):
        self._chk(self.L.svsdf_set_profiling(self.ctx, int(enable)), "svsdf_set_profiling")   # 2: serialised batches

    def stats(self):
        s = Stats()
        self.L.svsdf_last_stats(self.ctx, C.byref(s))
        return {k: getattr(s, k) for k, _ in Stats._fields_}
