#!/bin/bash
# round 4, run 36: Polygon with the one-formula cell lookup and the leaner record decode: C5 against r4head, the mesh
# outline tests, the Polygon SDF-at-time test, C5 through bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 200 python tools/exp_variants.py r4head C5 1000000 > gpurun_out/r4_36_c5_ab.txt 2>&1
cut -c1-300 gpurun_out/r4_36_c5_ab.txt
timeout 120 python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r4_36_bench_C5.json 2> gpurun_out/r4_36_bench_C5.err
python -c "
import json
b=json.loads(open('gpurun_out/r4_36_bench_C5.json').read().strip().splitlines()[-1]); print('bench C5', b['ms_per_step'], b['value'])"
timeout 400 python -m pytest tests/test_gpu_mesh_shapes.py tests/test_gpu_sdf_at.py -x -q -k "not sd or mesh or outline or Polygon" > gpurun_out/r4_36_pytest.txt 2>&1
grep -n "passed\|failed" gpurun_out/r4_36_pytest.txt | tail -3
