#!/bin/bash
# round 4, GPU call 8: fused GSIP tail with the in-flight threshold (4096), register cap 3 waves/SIMD (t1) vs none (t2)
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
for v in t1 t2; do
  timeout 600 python tools/ab_env.py $v "SVSDF_TAIL=off;SVSDF_TAIL=auto;SVSDF_TAIL=auto,SVSDF_TAIL_ALL_AFTER=-1;SVSDF_TAIL=auto,SVSDF_TAIL_BELOW=16384;SVSDF_TAIL=off,SVSDF_PIECE_TIME=exact" C1,C2,C3,NS 0 20 > gpurun_out/r4_8_tail_$v.txt 2>&1
  cat gpurun_out/r4_8_tail_$v.txt
done
