#!/bin/bash
# round 4, run 38: rocprofv3 kernel trace of 6 C5 evaluations at the final commit
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c5 -o c5 -- python $GRAFT_REPO_ROOT/tools/c5_driver.py C5 1000000 6 > $GRAFT_REPO_ROOT/gpurun_out/r4_38_kt.log 2>&1)
cp $(find /tmp/kt_c5 -name '*kernel_stats.csv' | head -1) gpurun_out/r04_v4_C5_kernel_stats.csv
head -12 gpurun_out/r04_v4_C5_kernel_stats.csv | cut -c1-160
