#!/bin/bash
# round 4, GPU call 9: full gpu test suite + default bench line at the stage-1 commit
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4_9_pytest.txt 2>&1
tail -15 gpurun_out/r4_9_pytest.txt
timeout 900 python bench.py > gpurun_out/r4_9_bench.json 2> gpurun_out/r4_9_bench.err
tail -c 600 gpurun_out/r4_9_bench.json; tail -5 gpurun_out/r4_9_bench.err
