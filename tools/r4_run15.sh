#!/bin/bash
# round 4, GPU call 15: where the fused tail stops paying: C1 workload at 3 k .. 30 k points, chain (SVSDF_TAIL=off) vs tail from iteration 0
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
rm -f gpurun_out/r4_15_tail_threshold.txt
for P in 3000 6000 10000 15000 20000 30000; do
  timeout 600 python tools/ab_env.py tw "SVSDF_TAIL=off;SVSDF_TAIL=0" C1 $P 20 >> gpurun_out/r4_15_tail_threshold.txt 2>&1
done
cut -c1-40,300-420 gpurun_out/r4_15_tail_threshold.txt
