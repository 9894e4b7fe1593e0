#!/bin/bash
# counters of one workload: tools/pmc.sh <lib variant or -> <config> <points> "<counters>" [env...]
V=$1; C=$2; P=$3; CTR=$4; shift 4
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
[ "$V" = "-" ] || export SVSDF_LIB_VARIANT=$V
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$V_$C
timeout 600 rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_${V}_$C -o p -- python $ROOT/tools/prof_eval.py $C $P 3 > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc_${V}_$C -name '*counter_collection.csv' | head -1)
python $ROOT/tools/pmc_agg.py $f
