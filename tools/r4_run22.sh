#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_sdf_at.py tests/test_gpu_plan.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
FUZZ_DEGENERATE=1 FUZZ_DEVICE_TRIG=1 python tools/fuzz_parity.py 40 457738 2>&1 | grep -E "^CASE|cases," | cut -c1-260
