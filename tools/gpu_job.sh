#!/bin/bash
# One parametrised driver for this project's GPU calls (round 5; replaces round 4's tools/r4_run*.sh, one file per call):
#   gpurun --timeout N -- 'bash tools/gpu_job.sh <tag> <step> [<step> ...]'
# Every step writes gpurun_out/<tag>_<step>*.  Steps (each bounded by its own timeout):
#   probe        tools/state_probe.py (workload timing vs process state)
#   refscale     bench.py --only reference_scale
#   reftrace     rocprofv3 kernel trace + per-launch timeline of a reference-scale callback (star map)
#   site:<lib>[:<configs>]   tools/site_stats.py of a -DSVSDF_SITE_STATS variant build (default C3,NS; ref:star = reference scale)
#   bench        bench.py default line (N = 1)
#   benchq       bench.py --no-extras --no-cpu-baseline (headline only)
#   benchv:<variant>   the same with libsvsdf_hip_<variant>.so (SVSDF_LIB_VARIANT)
#   abenv:<spec>[:<configs>[:<points>]]   tools/ab_env.py on the default library ("A=1;B=2": first spec = baseline)
#   stripes8     bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config C4 (8 stripes on this GPU)
#   ab:<variants>[:<configs>[:<points>]]   tools/exp_variants.py (variant builds vs the default library, identity hash)
#   pytest[:<expr>]   pytest -m gpu [-k expr]
#   fuzz:<cases>:<seed>   tools/fuzz_parity.py campaign
#   ktp:<config>:<points>[:<batches>]   the same at a given cloud size through tools/prof_eval.py (one stripe of a multi-GPU run)
#   kt:<config>  rocprofv3 --kernel-trace --stats of the bench command on one config, one batch (SVSDF_BATCHES=1) + timeline
#   pmc:<config> separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*) of the same command -> tools/pmc_summary.py
#   pmcl:<config>:<points>[:<batches>]   per-launch instruction counters of one evaluation (default: one batch)
#   pmcref[:<map>]  PMC passes (instruction mix, wave cycles, instruction cache) of reference-scale callbacks
#   profround:<config>   tools/profile_round.sh: bench line + kernel stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_*) + one-batch trace
#   smoke        __graft_entry__.smoke()
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
TAG=$1; shift
mkdir -p $OUT
cd $ROOT || exit 1
export TMPDIR=/tmp
for STEP in "$@"; do
  NAME=${STEP%%:*}; ARG=""; [ "$STEP" != "$NAME" ] && ARG=${STEP#*:}
  T0=$(date +%s)
  case $NAME in
    probe)    timeout 600 python -u tools/state_probe.py $(echo $ARG | tr ',' ' ') > $OUT/${TAG}_probe.txt 2>&1 ;;
    refscale) timeout 300 python -u bench.py --only reference_scale > $OUT/${TAG}_refscale.json 2> $OUT/${TAG}_refscale.err ;;
    reftrace)
      (cd /tmp && rm -rf /tmp/rt_$TAG && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt_$TAG -o kt -- python -u $ROOT/tools/ref_trace.py ${ARG:-star} 12 > $OUT/${TAG}_reftrace.log 2>&1
       KT=$(find /tmp/rt_$TAG -name '*kernel_trace.csv' | head -1); KS=$(find /tmp/rt_$TAG -name '*kernel_stats.csv' | head -1)
       [ -n "$KS" ] && cp $KS $OUT/${TAG}_reftrace_kernel_stats.csv
       [ -n "$KT" ] && python $ROOT/tools/timeline.py $KT 9 > $OUT/${TAG}_reftrace_timeline.txt 2>&1) ;;
    site)     IFS=: read -r V C <<< "$ARG"; timeout 400 python -u tools/site_stats.py ${V:-st} ${C:-C3,NS} > $OUT/${TAG}_site_${V:-st}.txt 2>&1 ;;
    bench)    timeout 900 python -u bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err ;;
    benchq)   timeout 300 python -u bench.py --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_quick.json 2> $OUT/${TAG}_bench_quick.err ;;
    benchv)   SVSDF_LIB_VARIANT=$ARG timeout 300 python -u bench.py --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_quick_$ARG.json 2> $OUT/${TAG}_bench_quick_$ARG.err ;;
    abenv)    IFS=: read -r SPEC C P <<< "$ARG"; timeout 600 python -u tools/ab_env.py - "$SPEC" ${C:-C3,NS} ${P:-0} > $OUT/${TAG}_abenv_$(echo ${C:-C3,NS} | tr ',' '_')_${P:-0}.txt 2>&1 ;;
    stripes8) timeout 600 python -u bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config C4 --steps 10 --no-extras > $OUT/${TAG}_stripes8.json 2> $OUT/${TAG}_stripes8.err ;;
    ab)       IFS=: read -r V C P <<< "$ARG"; timeout 900 python -u tools/exp_variants.py "$V" ${C:-C3,NS} ${P:-1000000} > $OUT/${TAG}_ab_$(echo $V | tr ',' '_')_$(echo ${C:-C3,NS} | tr ',' '_').txt 2>&1 ;;
    pytest)   if [ -n "$ARG" ]; then timeout 1500 python -m pytest tests -x -q -m gpu -k "$ARG" > $OUT/${TAG}_pytest.txt 2>&1; else timeout 1500 python -m pytest tests -x -q -m gpu --durations=20 > $OUT/${TAG}_pytest.txt 2>&1; fi; tail -3 $OUT/${TAG}_pytest.txt ;;
    fuzz)     IFS=: read -r N S <<< "$ARG"; timeout 900 python -u tools/fuzz_parity.py ${N:-100} ${S:-1} > $OUT/${TAG}_fuzz_${S:-1}.txt 2>&1 ;;
    kt)
      (cd /tmp && rm -rf /tmp/kt_$TAG && SVSDF_BATCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o kt -- python -u $ROOT/bench.py --config $ARG --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_kt_$ARG.log 2>&1
       KS=$(find /tmp/kt_$TAG -name '*kernel_stats.csv' | head -1); KT=$(find /tmp/kt_$TAG -name '*kernel_trace.csv' | head -1)
       [ -n "$KS" ] && cp $KS $OUT/${TAG}_bench_${ARG}_b1_kernel_stats.csv
       [ -n "$KT" ] && python $ROOT/tools/timeline.py $KT 6 > $OUT/${TAG}_bench_${ARG}_b1_timeline.txt 2>&1) ;;
    ktp)   # ktp:<config>:<points>[:<batches>]  kernel trace + per-launch timeline of one evaluation at a given cloud size (a multi-GPU stripe: C4:500000)
      IFS=: read -r C P B <<< "$ARG"
      (cd /tmp && rm -rf /tmp/ktp_$TAG && SVSDF_BATCHES=${B:-} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp_$TAG -o kt -- python -u $ROOT/tools/prof_eval.py $C $P 8 > $OUT/${TAG}_ktp_${C}_${P}_b${B:-auto}.log 2>&1
       KS=$(find /tmp/ktp_$TAG -name '*kernel_stats.csv' | head -1); KT=$(find /tmp/ktp_$TAG -name '*kernel_trace.csv' | head -1)
       [ -n "$KS" ] && cp $KS $OUT/${TAG}_${C}_${P}_b${B:-auto}_kernel_stats.csv
       [ -n "$KT" ] && python $ROOT/tools/timeline.py $KT 5 > $OUT/${TAG}_${C}_${P}_b${B:-auto}_timeline.txt 2>&1) ;;
    pmc)
      (cd /tmp && for CTR in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
         D=/tmp/pmc_${TAG}_$(echo $CTR | tr ' ' '_'); rm -rf $D
         timeout 400 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $D -o p -- python -u $ROOT/tools/prof_eval.py $ARG $(python -c "import sys; sys.path[:0]=['$ROOT/tools']; from svsdf_cfg import default_points; print(default_points('$ARG'))") 6 > /tmp/pmc.log 2>&1
       done
       python $ROOT/tools/pmc_agg.py $(find /tmp/pmc_${TAG}_* -name '*counter_collection.csv') > $OUT/${TAG}_pmc_$ARG.txt 2>&1) ;;
    pmcl)   # pmcl:<config>:<points>[:<batches>]  per-LAUNCH instruction counters of one evaluation (tools/pmc_per_launch.py)
      IFS=: read -r C P B <<< "$ARG"
      (cd /tmp && for CTR in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" ${PMCL_EXTRA:+"SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F64 SQ_INST_CYCLES_SALU SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_LEVEL_WAVES SQ_CYCLES"}; do
         D=/tmp/pmcl_${TAG}_$(echo $CTR | tr ' ' '_'); rm -rf $D
         SVSDF_BATCHES=${B:-1} timeout 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $D -o p -- python -u $ROOT/tools/prof_eval.py $C $P 6 > /tmp/pmcl.log 2>&1 || tail -3 /tmp/pmcl.log
         python $ROOT/tools/pmc_per_launch.py $(find $D -name '*counter_collection.csv' | head -1) 4 > $OUT/${TAG}_pmcl_${C}_${P}_b${B:-1}_$(echo $CTR | cut -d' ' -f1).txt 2>&1
       done) ;;
    pmcref)
      (cd /tmp && rocprofv3 -L > $OUT/${TAG}_counters_available.txt 2>&1
       for CTR in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INSTS_FLAT"; do
         D=/tmp/pmcref_${TAG}_$(echo $CTR | tr ' ' '_'); rm -rf $D
         timeout 200 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $D -o p -- python -u $ROOT/tools/ref_trace.py ${ARG:-star} 12 > /tmp/pmcref.log 2>&1 || tail -3 /tmp/pmcref.log
       done
       python $ROOT/tools/pmc_agg.py $(find /tmp/pmcref_${TAG}_* -name '*counter_collection.csv') > $OUT/${TAG}_pmcref_${ARG:-star}.txt 2>&1) ;;
    profround) bash tools/profile_round.sh ${TAG} ${ARG:-C3} > /dev/null 2>&1 ;;
    smoke)    timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.txt 2>&1; tail -2 $OUT/${TAG}_smoke.txt ;;
    *) echo "unknown step $STEP" ;;
  esac
  echo "[$TAG] $STEP rc=$? $(( $(date +%s) - T0 )) s"
done
