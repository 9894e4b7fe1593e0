set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python -u $ROOT/bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02_v4_kt -o kt -- $BENCH > $OUT/r02_v4_bench_C3_short.log 2>&1
KS=$(find $OUT/r02_v4_kt -name '*kernel_stats.csv' | head -1); cp $KS $OUT/r02_v4_bench_C3_kernel_stats.csv
KT=$(find $OUT/r02_v4_kt -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/timeline.py $KT 6 > $OUT/r02_v4_bench_C3_timeline.txt 2>&1
rm -rf $OUT/r02_v4_kt
grep '^{' $OUT/r02_v4_bench_C3_short.log | tail -1 > $OUT/r02_v4_bench_C3_short.json
timeout 100 python -u $ROOT/bench.py --config C2 --steps 50 --warmup 3 --no-extras --cpu-seconds 6 2>/dev/null | grep '^{' | tail -1 > $OUT/r02_v4_bench_C2.json
head -c 600 $OUT/r02_v4_bench_C2.json; echo; head -5 $OUT/r02_v4_bench_C3_kernel_stats.csv
