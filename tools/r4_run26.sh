#!/bin/bash
# round 4, GPU call 26: plan / tail / anchor tests, then C3 / C4 / C5 / NS timing at this build
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_properties.py tests/test_gpu_exactness_all_shapes.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
for c in C3 C4 C5 NS; do python tools/scan_counts.py - $c 1000000 2>&1 | tail -1; done
