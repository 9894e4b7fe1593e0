#!/bin/bash
# Everything a round is judged on, on the GPU box (through gpurun):  build check, GPU tests, smoke, bench,
# rocprof evidence.   usage: gpurun --timeout 1500 -- 'bash tools/run_all.sh r01_vN'
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${TAG}_build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/${TAG}_build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.log
bash tools/profile_round.sh $TAG > /dev/null
tail -c 400 gpurun_out/${TAG}_bench_C2.json
