#!/bin/bash
# round 4, run 34: the final Polygon evaluation (three grid levels, bisector-refined lists, wave walk, cell parity) against
# the round's HEAD (variant r4head = fast build of commit 11ec2d6), then the whole GPU suite and the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/exp_variants.py r4head,polyF C5 1000000 > gpurun_out/r4_34_c5_ab.txt 2>&1
timeout 400 python tools/poly_outline_ab.py r4head star,sdRoundedCross,sdArc 200000 5 > gpurun_out/r4_34_outline_ab.txt 2>&1
cat gpurun_out/r4_34_c5_ab.txt gpurun_out/r4_34_outline_ab.txt | cut -c1-300
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_34_pytest.txt 2>&1
tail -3 gpurun_out/r4_34_pytest.txt
timeout 400 python bench.py > gpurun_out/r4_34_bench.json 2> gpurun_out/r4_34_bench.err
python - <<'P'
import json
b=json.loads(open('gpurun_out/r4_34_bench.json').read().strip().splitlines()[-1])
print('headline', round(b['ms_per_step'],3), 'NS', round(b['north_star']['ms_per_step'],3), 'C5', b['other_configs']['C5']['ms_per_step'], b['other_configs']['C5'].get('generic_durations_ms_per_step'), 'C1', b['other_configs']['C1']['ms_per_step'], 'C2', b['other_configs']['C2']['ms_per_step'], 'c4', b['c4_one_gpu']['ms_per_step'])
P
