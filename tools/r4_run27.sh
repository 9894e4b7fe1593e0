#!/bin/bash
# round 4, GPU call 27: gpu tests + default bench line + rocprofv3 evidence (C3) at this commit
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4_27_pytest.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4_27_pytest.txt | tail -4
timeout 900 python bench.py > gpurun_out/r4_27_bench.json 2> gpurun_out/r4_27_bench.err
tail -c 300 gpurun_out/r4_27_bench.json
bash tools/profile_round.sh r04_v2 C3 > gpurun_out/r4_27_profile_C3.log 2>&1
tail -3 gpurun_out/r4_27_profile_C3.log
bash tools/profile_round.sh r04_v2 NS > gpurun_out/r4_27_profile_NS.log 2>&1
tail -2 gpurun_out/r4_27_profile_NS.log | cut -c1-200
