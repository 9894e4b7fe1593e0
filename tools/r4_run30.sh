#!/bin/bash
# round 4, run 30: where does a Polygon evaluation wait?  PMC passes of 6 C5 evaluations, default build and polyD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for V in default polyD; do
  if [ $V = default ]; then unset SVSDF_LIB_VARIANT; else export SVSDF_LIB_VARIANT=$V; fi
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pm_${V}_a -o a -- python $GRAFT_REPO_ROOT/tools/c5_driver.py C5 1000000 6 > $O/r4_30_${V}_a.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pm_${V}_b -o b -- python $GRAFT_REPO_ROOT/tools/c5_driver.py C5 1000000 6 > $O/r4_30_${V}_b.log 2>&1)
  python tools/pmc_agg.py $(find /tmp/pm_${V}_a /tmp/pm_${V}_b -name '*counter_collection.csv') > $O/r4_30_pmc_${V}.txt 2>&1
done
head -14 $O/r4_30_pmc_default.txt | cut -c1-260
