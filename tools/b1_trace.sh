set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; TAG=r02_v3
cd /tmp && export TMPDIR=/tmp
for CFG in C3 NS; do
BENCH="python -u $ROOT/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
SVSDF_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt1 -o kt -- $BENCH > $OUT/${TAG}_kt1.log 2>&1
KS1=$(find $OUT/${TAG}_kt1 -name '*kernel_stats.csv' | head -1); cp $KS1 $OUT/${TAG}_bench_${CFG}_b1_kernel_stats.csv
KT1=$(find $OUT/${TAG}_kt1 -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/timeline.py $KT1 6 > $OUT/${TAG}_bench_${CFG}_b1_timeline.txt 2>&1
rm -rf $OUT/${TAG}_kt1
done
tail -14 $OUT/${TAG}_bench_C3_b1_timeline.txt
