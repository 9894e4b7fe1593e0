#!/bin/bash
# per-launch timeline of one evaluation: tools/kt.sh <lib variant or -> <config> [points] [extra env...]
V=$1; C=$2; P=${3:-1000000}; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
[ "$V" = "-" ] || export SVSDF_LIB_VARIANT=$V
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$V_$C
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_${V}_$C -o kt -- python $ROOT/tools/prof_eval.py $C $P 5 > /tmp/kt.log 2>&1
f=$(find /tmp/kt_${V}_$C -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/timeline.py $f 3 | tee $OUT/timeline_${V}_${C}.txt | tail -22
