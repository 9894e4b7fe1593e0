#!/bin/bash
# round 4, run 37: smoke + the default bench line at the final commit
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_37_smoke.txt 2>&1; tail -1 gpurun_out/r4_37_smoke.txt
timeout 230 python bench.py > gpurun_out/r4_37_bench.json 2> gpurun_out/r4_37_bench.err
python - <<'P'
import json
b=json.loads(open('gpurun_out/r4_37_bench.json').read().strip().splitlines()[-1])
print('headline', round(b['ms_per_step'],3), round(b['value']/1e6,1), 'generic', round(b['generic_durations']['ms_per_step'],3), 'NS', round(b['north_star']['ms_per_step'],3), 'C5', b['other_configs']['C5']['ms_per_step'], b['other_configs']['C5'].get('generic_durations_ms_per_step'), 'C1', b['other_configs']['C1']['ms_per_step'], 'C2', b['other_configs']['C2']['ms_per_step'], 'c4', b['c4_one_gpu']['ms_per_step'], 'cpu', b['cpu_baseline']['value'])
P
