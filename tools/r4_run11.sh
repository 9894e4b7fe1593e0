#!/bin/bash
# round 4, GPU call 11: lean GSIP hand-off (records for requested samples only, compact interior index) vs round 3: identity + time
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
rm -f gpurun_out/r4_11_handoff.txt
for c in C1:10000 C3:1000000 NS:1000000 C3:1000000 NS:1000000; do
  timeout 600 python tools/exp_variants.py base,h2,h3 ${c%%:*} ${c##*:} >> gpurun_out/r4_11_handoff.txt 2>&1
done
cat gpurun_out/r4_11_handoff.txt | cut -c1-330
