#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + PMC passes of the same command.
# usage: tools/profile_round.sh <tag> [config]   -> writes gpurun_out/<tag>_*   (config: C3 default, NS, C2 ...)
set -u
TAG=${1:-r02}
CFG=${2:-C3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python -u $ROOT/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
timeout 600 $BENCH > $OUT/${TAG}_bench_${CFG}_short.json 2> $OUT/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- $BENCH > $OUT/${TAG}_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_f -o f -- $BENCH > $OUT/${TAG}_pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_w -o w -- $BENCH > $OUT/${TAG}_pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_s -o s -- $BENCH > $OUT/${TAG}_pmc_s.log 2>&1
# summaries small enough to come back; the raw traces stay on the box
KS=$(find $OUT/${TAG}_kt -name '*kernel_stats.csv' | head -1); cp $KS $OUT/${TAG}_bench_${CFG}_kernel_stats.csv
F=$(find $OUT/${TAG}_pmc_f -name '*counter_collection.csv' | head -1)
W=$(find $OUT/${TAG}_pmc_w -name '*counter_collection.csv' | head -1)
S=$(find $OUT/${TAG}_pmc_s -name '*counter_collection.csv' | head -1)
# evaluations per profiled run: bench.py counts them itself (settle + warm-up + timed + event-profiled + serialised-profiled +
# generic-durations + full-callback passes)
NEV=$(python -c "import json,sys; print(json.load(open('$OUT/${TAG}_bench_${CFG}_short.json'))['evaluations_in_this_run'])")
python $ROOT/tools/pmc_summary.py $F $W $S ${3:-$NEV} "rocprofv3 PMC summary, bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-extras, 1x MI355X, $TAG" > $OUT/${TAG}_bench_${CFG}_pmc_summary.txt
KT=$(find $OUT/${TAG}_kt -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/timeline.py $KT 6 > $OUT/${TAG}_bench_${CFG}_timeline.txt 2>&1
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w $OUT/${TAG}_pmc_s
tail -c 300 $OUT/${TAG}_bench_${CFG}_short.json
# the same kernel trace with the point batches run one after the other: per-launch durations are then each kernel's own cost
SVSDF_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt1 -o kt -- $BENCH > $OUT/${TAG}_kt1.log 2>&1
KS1=$(find $OUT/${TAG}_kt1 -name '*kernel_stats.csv' | head -1); cp $KS1 $OUT/${TAG}_bench_${CFG}_b1_kernel_stats.csv
KT1=$(find $OUT/${TAG}_kt1 -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/timeline.py $KT1 6 > $OUT/${TAG}_bench_${CFG}_b1_timeline.txt 2>&1
rm -rf $OUT/${TAG}_kt1
