#!/bin/bash
# round 4, GPU call 24: fuzz against the oracle of record with the 1-ulp sensitivity bracket, two fresh seeds x 100 cases
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8 FUZZ_DEGENERATE=1
rm -f gpurun_out/r4_24_fuzz_bracket.txt
for seed in 161803 577215; do
  python tools/fuzz_parity.py 100 $seed 2>&1 | grep -E "^CASE|cases," | cut -c1-420 >> gpurun_out/r4_24_fuzz_bracket.txt
done
cat gpurun_out/r4_24_fuzz_bracket.txt
