#!/bin/bash
# round 4, GPU call 5: instruction counters of k_solve, fast vs faithful piece time (what the +20 % consists of)
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
CTR="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"
for v in c3 r3; do
  echo "== $v fast"; bash tools/pmc.sh $v C3 1000000 "$CTR" SVSDF_TAIL=off SVSDF_UB_FULL=1 SVSDF_BATCHES=1 2>&1 | grep -v "^$" | head -8
  echo "== $v exact"; bash tools/pmc.sh $v C3 1000000 "$CTR" SVSDF_TAIL=off SVSDF_UB_FULL=1 SVSDF_BATCHES=1 SVSDF_PIECE_TIME=exact 2>&1 | grep -v "^$" | head -8
done > gpurun_out/r4_5_pmc_chain.txt 2>&1
cat gpurun_out/r4_5_pmc_chain.txt
tail -5 /tmp/pmc.log
