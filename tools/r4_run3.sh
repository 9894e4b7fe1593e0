#!/bin/bash
# round 4, GPU call 4: chain v4 (operands of the chain end requested before the chain)
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
for v in c2n c2 r3; do
  timeout 300 python tools/ab_env.py $v "SVSDF_TAIL=off;SVSDF_TAIL=off,SVSDF_PIECE_TIME=exact" C3,NS 0 10 > gpurun_out/r4_7_chain_$v.txt 2>&1
  cat gpurun_out/r4_7_chain_$v.txt
done
