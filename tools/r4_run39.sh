#!/bin/bash
# round 4, run 39: the remaining Polygon tests at the final commit
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 150 python -m pytest -x -q -m gpu tests/test_frontend.py tests/test_golden_vectors.py tests/test_gpu_exactness_all_shapes.py tests/test_gpu_parity.py tests/test_gpu_plan.py tests/test_gpu_properties.py -k "Polygon or C5" > gpurun_out/r4_39_pytest.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/r4_39_pytest.txt | tail -3
