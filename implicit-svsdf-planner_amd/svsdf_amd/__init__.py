"""svsdf_amd -- Python host mirror of the reference's TrajOptimizer interface for the SVSDF
safety cost/gradient path, on top of the C ABI (include/svsdf_c.h) of the gfx950 library.

PyTorch is used only for device memory and torch.distributed (RCCL) plumbing; the compute is
in csrc/ (hand-written HIP).  There is no CPU fallback: every compute call raises if the HIP
library or a GPU is missing.
"""
from .binding import (SHAPES, SHAPE_ID, SvsdfError, SvsdfContext, lib, lib_path, minco_coeffs,
                      forward_T, backward_T, shape_id_from_inputdata, shard_plan,
                      FLAG_HOST_ONLY, FLAG_KEEP_INPUT_ORDER, FLAG_EXACT_PIECE_TIME, FLAG_FAST_PIECE_TIME, OccupancyMap, lbfgs_minimize, lbfgs_params,
                      LBFGS_STATUS, sum_partials, COMBINE_AUTO, COMBINE_HOST, COMBINE_RCCL, mesh_outline, mesh_outline_obj, outline_extrude, mesh_section, mesh_section_obj)
from .traj_optimizer import TrajOptimizer
from . import workload

__all__ = ["SHAPES", "SHAPE_ID", "SvsdfError", "SvsdfContext", "TrajOptimizer", "lib", "lib_path",
           "minco_coeffs", "forward_T", "backward_T", "shape_id_from_inputdata", "shard_plan",
           "FLAG_HOST_ONLY", "FLAG_KEEP_INPUT_ORDER", "FLAG_EXACT_PIECE_TIME", "FLAG_FAST_PIECE_TIME", "OccupancyMap", "workload", "lbfgs_minimize",
           "lbfgs_params", "LBFGS_STATUS", "sum_partials", "COMBINE_AUTO", "COMBINE_HOST", "COMBINE_RCCL",
           "mesh_outline", "mesh_outline_obj", "outline_extrude", "mesh_section", "mesh_section_obj"]
