#!/bin/bash
# round 4, run 32: Polygon (C5) PMC -- instruction mix, lane utilisation and instruction-cache behaviour, default vs polyE
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for V in default polyE; do
  if [ $V = default ]; then unset SVSDF_LIB_VARIANT; else export SVSDF_LIB_VARIANT=$V; fi
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pn_${V}_a -o a -- python $GRAFT_REPO_ROOT/tools/c5_driver.py C5 1000000 6 > $O/r4_32_${V}_a.log 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/pn_${V}_b -o b -- python $GRAFT_REPO_ROOT/tools/c5_driver.py C5 1000000 6 > $O/r4_32_${V}_b.log 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pn_${V}_c -o c -- python $GRAFT_REPO_ROOT/tools/c5_driver.py C5 1000000 6 > $O/r4_32_${V}_c.log 2>&1)
  python tools/pmc_agg.py $(find /tmp/pn_${V}_a /tmp/pn_${V}_b /tmp/pn_${V}_c -name '*counter_collection.csv') > $O/r4_32_pmc_${V}.txt 2>&1
done
# the same for NS (analytic star) as the yardstick
unset SVSDF_LIB_VARIANT
(cd /tmp && timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pn_ns_c -o c -- python $GRAFT_REPO_ROOT/tools/c5_driver.py NS 1000000 6 > $O/r4_32_ns_c.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pn_ns_a -o a -- python $GRAFT_REPO_ROOT/tools/c5_driver.py NS 1000000 6 > $O/r4_32_ns_a.log 2>&1)
python tools/pmc_agg.py $(find /tmp/pn_ns_a /tmp/pn_ns_c -name '*counter_collection.csv') > $O/r4_32_pmc_ns.txt 2>&1
grep -h "k_solve<17, 4\|kernel  \|^#" $O/r4_32_pmc_default.txt $O/r4_32_pmc_polyE.txt | cut -c1-240
