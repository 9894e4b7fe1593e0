"""Host-side mirror of the reference's TrajOptimizer for the SVSDF cost path.

Same member / method names, argument meaning and error behaviour as
src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp (BEO) for the functions
on the hot path:
    TrajOptimizer::setParam                                   BEO:872-935
    TrajOptimizer::costFunctionLmbmParallel(ptr, x, g, n)     BEO:344-408
    TrajOptimizer::addSaftyPenaOnSweptVolumeParallelTrueSDF   BEO:774-869
    cost_pos / cost_other / cost_total                        BEO:396-398
The compute runs in libsvsdf_hip.so (gfx950); this class only marshals arguments.  With
torch.distributed initialised (one process per GPU) the query points are sharded over the ranks
and the (19N+1)-double partial is summed with one all-reduce (RCCL on GPU ranks).
"""
import os

import numpy as np

from .binding import SvsdfContext, SvsdfError, shape_id_from_inputdata, SHAPES, mesh_section_obj


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


class TrajOptimizer:
    def __init__(self):
        # Config-derived members (BEO:69-95)
        self.rho = 3.8
        self.weight_p = 60.0
        self.safety_hor = 0.7
        self.threads_num = 30          # kept for interface compatibility; unused on the GPU
        self.inputdata = "shapes/star.obj"
        self.poly_params = (0.0, 0.0, 0.0)
        self.polygon = None
        self.polygon_loops = None   # with `polygon`: vertex counts of its closed loops (None: one loop)
        self.package_path = ""         # what ros::package::getPath("plan_manager") returns (Shape.hpp:283)
        self.parallel_points = np.zeros((0, 3))
        self.parallel_points_num = 0
        self.cost_pos = 0.0
        self.cost_other = 0.0
        self.cost_total = 0.0
        self.pieceN = 0
        self.temporalDim = 0
        self.spatialDim = 0
        self.initState = np.zeros((3, 3))
        self.finalState = np.zeros((3, 3))
        self.device = -1
        self.devices = None            # list of >= 2 HIP ordinals: in-process multi-GPU context
        self.combine = 0
        self._ctx = None
        self._points_dirty = True

    # -- TrajOptimizer::setParam (reads Config; here a dict with the yaml keys) --
    def setParam(self, config):
        self.rho = float(config.get("rho", self.rho))
        self.weight_p = float(config.get("weight_p", self.weight_p))
        self.safety_hor = float(config.get("safety_hor", self.safety_hor))
        self.threads_num = int(config.get("threads_num", self.threads_num))
        self.inputdata = config.get("inputdata", self.inputdata)
        self.poly_params = tuple(config.get("poly_params", self.poly_params))
        self.polygon = config.get("polygon", self.polygon)
        self.polygon_loops = config.get("polygon_loops", self.polygon_loops)
        self.package_path = config.get("package_path", self.package_path)
        self.device = int(config.get("device", self.device))
        self.devices = config.get("devices", self.devices)
        self.combine = int(config.get("combine", self.combine))
        self._ctx = None

    def setPoints(self, xyz):
        """plan_manager.cpp:168-175: parallel_points / parallel_points_num."""
        self.parallel_points = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        self.parallel_points_num = len(self.parallel_points)
        self._points_dirty = True

    # -- first lines of optimize_traj_lmbm (back_end_optimizer.cpp:13-19) --
    def setConditions(self, initS, finalS, N):
        self.pieceN = int(N)
        self.temporalDim = int(N)
        self.spatialDim = 3 * (int(N) - 1)
        self.initState = np.asarray(initS, dtype=np.float64).reshape(3, 3)
        self.finalState = np.asarray(finalS, dtype=np.float64).reshape(3, 3)
        # the context (resident cloud, launch plan, persistent traj_duration SWM:376-385) survives
        # successive optimisations; only the boundary states change
        if self._ctx is not None:
            self._ctx.set_conditions(self.initState, self.finalState)

    def _context(self):
        if self._ctx is None:
            sid = shape_id_from_inputdata(self.inputdata) if self.polygon is None else SHAPES.index("Polygon")
            polygon = self.polygon
            polygon_loops = getattr(self, "polygon_loops", None)
            if sid == SHAPES.index("Polygon") and polygon is None:
                # an .obj the shape registry does not know (BASELINE config 5): its z = 0 outline; unreadable -> the
                # reference's hard-coded rectangle (sw_manager.hpp:363-369), which the library substitutes itself
                path = os.path.join(self.package_path, self.inputdata) if self.package_path else self.inputdata
                if os.path.exists(path):
                    # the whole section: every closed loop (a hole, several solids; round 5 -- rounds 3-4 refused them)
                    polygon, polygon_loops = mesh_section_obj(path)
            dist = _dist()
            rank, ws = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
            self._ctx = SvsdfContext(shape=sid, safety_hor=self.safety_hor, weight_p=self.weight_p,
                                     rho=self.rho, poly_params=self.poly_params, polygon=polygon,
                                     head_state=self.initState, tail_state=self.finalState,
                                     device=self.device, rank=rank, world_size=ws,
                                     devices=self.devices, combine=self.combine, polygon_loops=polygon_loops)
            self._points_dirty = True
        if self._points_dirty:
            self._ctx.set_points(self.parallel_points)
            self._points_dirty = False
        return self._ctx

    def _allreduce_partial(self, ptr, n):
        """Sum the device-resident partial over the ranks (RCCL) and return it on the host."""
        import torch
        dist = _dist()
        dev = torch.device("cuda", torch.cuda.current_device())
        # wrap the library's device buffer without copying
        t = _wrap_device_f64(ptr, n, dev)
        if dist is not None and dist.get_backend() != "nccl":
            # host-side collective (gloo): used to emulate several ranks on one GPU in tests; RCCL is the product path
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            return h.numpy()
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    # -- static double costFunctionLmbmParallel(void* ptr, const double* x, double* g, int n) --
    def costFunctionLmbmParallel(self, x, g=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = len(x)
        if n != self.temporalDim + self.spatialDim:
            raise SvsdfError("n != temporalDim + spatialDim")
        ctx = self._context()
        dist = _dist()
        if dist is not None:   # one process per GPU: shard partials are summed with one all-reduce
            ptr, plen = ctx.lmbm_begin(x)
            partial = self._allreduce_partial(ptr, plen)
            cost, grad = ctx.lmbm_finish(partial, n)
        else:
            cost, grad = ctx.lmbm_evaluate(x)
        self.cost_pos, self.cost_other, self.cost_total = ctx.last_costs()
        if g is not None:
            g[:] = grad
        return cost, grad

    # -- static void addSaftyPenaOnSweptVolumeParallelTrueSDF(ptr, T, coeffs, cost, gradT, gradC) --
    def addSaftyPenaOnSweptVolumeParallelTrueSDF(self, T, coeffs, cost, gradT, gradC):
        """Accumulates into (cost, gradT, gradC) like the reference; returns the updated triple."""
        ctx = self._context()
        dist = _dist()
        if dist is not None:
            ptr, plen = ctx.eval_penalty_partial(coeffs, T)
            partial = self._allreduce_partial(ptr, plen)
            return ctx.accumulate_partial(len(T), partial, cost, gradT, gradC)
        return ctx.eval_penalty(coeffs, T, cost, gradT, gradC)

    def stats(self):
        return self._context().stats()


def _wrap_device_f64(ptr, n, device):
    """torch tensor view over `n` doubles at device pointer `ptr` (no copy, no ownership)."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)
