#!/bin/bash
# round 4, GPU call 14: k_tail with one point per wave + 32-lane groups (tw) vs default, small clouds
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
rm -f gpurun_out/r4_14_tail_small.txt
for c in C1:3000 C1:10000 C1:30000 C2:30000 C1:10000; do
  timeout 600 python tools/exp_variants.py tw ${c%%:*} ${c##*:} >> gpurun_out/r4_14_tail_small.txt 2>&1
done
cat gpurun_out/r4_14_tail_small.txt | cut -c1-60,280-420
