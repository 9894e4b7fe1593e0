#!/bin/bash
# round 4, run 28: Polygon (C5) -- refined candidate lists (polyA), + wave-walk / reciprocal quotient / cell parity (polyB)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python tools/exp_variants.py polyA,polyB C5 1000000 > gpurun_out/r4_28_ab.txt 2>&1
SVSDF_LIB_VARIANT=polyB timeout 500 python -m pytest tests/test_gpu_mesh_shapes.py -x -q > gpurun_out/r4_28_mesh_polyB.txt 2>&1
tail -30 gpurun_out/r4_28_ab.txt; tail -5 gpurun_out/r4_28_mesh_polyB.txt
