#!/bin/bash
# round 4, GPU call 13: split host layer + deferred (value, index) reduction in the table scans vs the previous commit ("base")
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
rm -f gpurun_out/r4_13_ab.txt
for c in C1:10000 C2:100000 C3:1000000 NS:1000000 C5:300000 C3:1000000 NS:1000000; do
  timeout 600 python tools/exp_variants.py base ${c%%:*} ${c##*:} >> gpurun_out/r4_13_ab.txt 2>&1
done
cat gpurun_out/r4_13_ab.txt | cut -c1-60,300-420
