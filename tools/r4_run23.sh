#!/bin/bash
# round 4, GPU call 23: device-arithmetic fuzz campaign on fresh seeds (bit-identity), 6 x 60 cases
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8 FUZZ_DEGENERATE=1 FUZZ_DEVICE_TRIG=1
rm -f gpurun_out/r4_23_fuzz_devtrig.txt
for seed in 31337 90210 271828 314159 8675309 112358; do
  python tools/fuzz_parity.py 60 $seed 2>&1 | grep -E "^CASE|cases," | cut -c1-300 >> gpurun_out/r4_23_fuzz_devtrig.txt
done
cat gpurun_out/r4_23_fuzz_devtrig.txt
