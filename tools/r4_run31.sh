#!/bin/bash
# round 4, run 31: Polygon (C5) -- polyE = polyD + the far grid level (no query walks all edges any more)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python tools/exp_variants.py polyE C5 1000000 > gpurun_out/r4_31_ab.txt 2>&1
SVSDF_LIB_VARIANT=polyE timeout 500 python -m pytest tests/test_gpu_mesh_shapes.py -x -q > gpurun_out/r4_31_mesh_polyE.txt 2>&1
tail -30 gpurun_out/r4_31_ab.txt; tail -5 gpurun_out/r4_31_mesh_polyE.txt
