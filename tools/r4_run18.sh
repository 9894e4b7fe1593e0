#!/bin/bash
# round 4, GPU call 18: late fused tail re-measured after the fetch-granularity fix: SVSDF_TAIL = off / 5..10 at NS, C3, C2
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
rm -f gpurun_out/r4_18_late_tail.txt
timeout 900 python tools/ab_env.py - "SVSDF_TAIL=off;SVSDF_TAIL=5;SVSDF_TAIL=6;SVSDF_TAIL=7;SVSDF_TAIL=8;SVSDF_TAIL=9;SVSDF_TAIL=10" NS,C3 1000000 10 >> gpurun_out/r4_18_late_tail.txt 2>&1
timeout 900 python tools/ab_env.py - "SVSDF_TAIL=off;SVSDF_TAIL=3;SVSDF_TAIL=4;SVSDF_TAIL=5;SVSDF_TAIL=6;SVSDF_TAIL=7" C2 100000 20 >> gpurun_out/r4_18_late_tail.txt 2>&1
python - <<'PY'
import json,re
for l in open('gpurun_out/r4_18_late_tail.txt'):
    m=re.search(r'^(\w+) +\[(.*?)\] (\{.*\}) identical=(\w+)',l)
    if m:
        d=json.loads(m.group(3)); print(m.group(1), m.group(2), round(d['ms'],3), m.group(4))
PY
