#!/bin/bash
# round-2 baseline of the round-1 build on the 1 M-point workloads: bench lines + per-launch timelines
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python -u $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
SVSDF_UB_FULL=1 timeout 300 $B --config C3 > $OUT/r02_base_C3_full.json 2> $OUT/r02_base.err
SVSDF_UB_FULL=1 timeout 300 $B --config C2 --points 1000000 > $OUT/r02_base_NS_full.json 2>> $OUT/r02_base.err
SVSDF_UB_FULL=0 timeout 300 $B --config C2 --points 1000000 > $OUT/r02_base_NS_cheap.json 2>> $OUT/r02_base.err
for m in C3:1:C3 NS:1:C2 NS:0:C2; do
  IFS=: read name ub cfg <<< "$m"
  extra=""; [ $name = NS ] && extra="--points 1000000"
  SVSDF_UB_FULL=$ub timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r02_base_kt_${name}_$ub -o kt -- python -u $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --config $cfg $extra > $OUT/r02_base_kt_${name}_$ub.log 2>&1
  f=$(find $OUT/r02_base_kt_${name}_$ub -name '*kernel_trace.csv' | head -1)
  python $ROOT/tools/timeline.py $f 8 > $OUT/r02_base_timeline_${name}_$ub.txt 2>&1
  rm -rf $OUT/r02_base_kt_${name}_$ub
done
tail -c 300 $OUT/r02_base_C3_full.json
