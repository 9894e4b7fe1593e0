#!/bin/bash
# round 4, run 35: Polygon on large outlines, HEAD-before-the-rewrite (r4head) against the final code
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/poly_outline_ab.py r4head sdRoundedCross,sdArc 200000 5 > gpurun_out/r4_35_outline_ab.txt 2>&1
cut -c1-400 gpurun_out/r4_35_outline_ab.txt
