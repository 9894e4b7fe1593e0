#!/bin/bash
# round 4, run 33: Polygon (C5) -- lanes per query 2 / 4 / 8 on the default build and on polyE
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/ab_env.py - ";SVSDF_G=2;SVSDF_G=8;SVSDF_BATCHES=1" C5 1000000 10 > gpurun_out/r4_33_lanes_default.txt 2>&1
timeout 200 python tools/ab_env.py polyE ";SVSDF_G=2" C5 1000000 10 > gpurun_out/r4_33_lanes_polyE.txt 2>&1
cat gpurun_out/r4_33_lanes_default.txt gpurun_out/r4_33_lanes_polyE.txt | cut -c1-330
