#!/bin/bash
# HBM traffic of one workload for the default library and a variant: tools/traffic_ab.sh <config> <variant>
C=${1:-C3}; V=${2:-pointmajor}   # e.g. python build.py --variant pointmajor -DSVSDF_POINT_MAJOR
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "" $V; do
  export SVSDF_LIB_VARIANT=$L
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tab_$CTR
    timeout 300 rocprofv3 --pmc $CTR --output-format csv -d /tmp/tab_$CTR -o p -- python $ROOT/tools/prof_eval.py $C 1000000 6 > /tmp/tab.log 2>&1
    f=$(find /tmp/tab_$CTR -name '*counter_collection.csv' | head -1)
    echo "== lib '${L:-default}' $CTR (6 evaluations + setup)"; python $ROOT/tools/pmc_agg.py $f | cut -c1-80 | head -6
  done
done
