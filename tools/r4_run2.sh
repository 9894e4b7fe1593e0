#!/bin/bash
# round 4, GPU call 2: chain operand sources (scalar window / LDS window / lane table) and the fused GSIP tail
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
for v in c3 c2 c4; do
  timeout 300 python tools/ab_env.py $v "SVSDF_TAIL=off;SVSDF_TAIL=off,SVSDF_PIECE_TIME=exact" C3,NS 0 10 > gpurun_out/r4_2_chain_$v.txt 2>&1
  cat gpurun_out/r4_2_chain_$v.txt
done
timeout 600 python tools/ab_env.py c3 "SVSDF_TAIL=off;SVSDF_TAIL=auto;SVSDF_TAIL=2;SVSDF_TAIL=3;SVSDF_TAIL=4;SVSDF_TAIL=5;SVSDF_TAIL=6" C3,NS 0 10 > gpurun_out/r4_2_tail_1M.txt 2>&1
cat gpurun_out/r4_2_tail_1M.txt
timeout 600 python tools/ab_env.py c3 "SVSDF_TAIL=off;SVSDF_TAIL=auto;SVSDF_TAIL=0;SVSDF_TAIL=1;SVSDF_TAIL=2;SVSDF_TAIL=3" C2,C1 0 20 > gpurun_out/r4_2_tail_small.txt 2>&1
cat gpurun_out/r4_2_tail_small.txt
