#!/bin/bash
# round 4, run 29: Polygon (C5) -- polyC: parity record fetched only when the cell needs it; polyD: + the cell's own
# crossing list packed into its record (one gather per evaluation)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python tools/exp_variants.py polyB,polyC,polyD C5 1000000 > gpurun_out/r4_29_ab.txt 2>&1
SVSDF_LIB_VARIANT=polyD timeout 500 python -m pytest tests/test_gpu_mesh_shapes.py -x -q > gpurun_out/r4_29_mesh_polyD.txt 2>&1
tail -30 gpurun_out/r4_29_ab.txt; tail -5 gpurun_out/r4_29_mesh_polyD.txt
