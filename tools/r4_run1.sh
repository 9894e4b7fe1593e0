#!/bin/bash
# round 4, GPU call 1: new piece-time chain (scalar-loaded durations) vs round 3
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "exact_piece_time or device_arithmetic or bit_identical" > gpurun_out/r4_1_pytest.txt 2>&1
tail -3 gpurun_out/r4_1_pytest.txt
timeout 600 python tools/ab_env.py r3 ";SVSDF_PIECE_TIME=exact" C3,NS 0 10 > gpurun_out/r4_1_ab_r3.txt 2>&1
timeout 600 python tools/ab_env.py pt1 ";SVSDF_PIECE_TIME=exact" C3,NS 0 10 > gpurun_out/r4_1_ab_pt1.txt 2>&1
cat gpurun_out/r4_1_ab_r3.txt gpurun_out/r4_1_ab_pt1.txt
