#!/bin/bash
# round 4, GPU call 17: lanes per query (SVSDF_G) against the shard size after the shared ladders: C2 workload 20 k .. 300 k points
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
rm -f gpurun_out/r4_17_lanes.txt
for P in 20000 50000 100000 200000 300000; do
  timeout 600 python tools/ab_env.py - ";SVSDF_G=2;SVSDF_G=4;SVSDF_G=8;SVSDF_G=16" C2 $P 20 >> gpurun_out/r4_17_lanes.txt 2>&1
done
python - <<'PY'
import json,re
for l in open('gpurun_out/r4_17_lanes.txt'):
    m=re.search(r'\[(.*?)\] (\{.*\}) identical=(\w+)',l)
    if m:
        d=json.loads(m.group(2)); print(m.group(1) or 'default', round(d['ms'],3), d['solves'], m.group(3))
PY
bash tools/profile_round.sh r04_v1 NS > gpurun_out/r4_17_profile_NS.log 2>&1
tail -2 gpurun_out/r4_17_profile_NS.log | cut -c1-200
