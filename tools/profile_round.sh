#!/bin/bash
# Run on the GPU box (through gpurun): bench + rocprofv3 kernel stats + PMC passes of the same command.
# usage: tools/profile_round.sh <tag>   -> writes gpurun_out/<tag>_*
set -u
TAG=${1:-r01_v4}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python -u $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 python -u $ROOT/bench.py > $OUT/${TAG}_bench_C2.json 2> $OUT/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- $BENCH > $OUT/${TAG}_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_f -o f -- $BENCH > $OUT/${TAG}_pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_w -o w -- $BENCH > $OUT/${TAG}_pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_s -o s -- $BENCH > $OUT/${TAG}_pmc_s.log 2>&1
find $OUT -name '*.csv' | head -30
