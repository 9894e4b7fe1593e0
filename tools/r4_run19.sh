#!/bin/bash
# round 4, GPU call 19: second (value-based) exact cull vs the committed build
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
rm -f gpurun_out/r4_19_cull2.txt
for c in C3:1000000 NS:1000000 C4:1000000 C2:100000 C5:300000 C1:10000 C3:1000000 NS:1000000; do
  timeout 600 python tools/exp_variants.py cu2 ${c%%:*} ${c##*:} >> gpurun_out/r4_19_cull2.txt 2>&1
done
python - <<'PY'
import json,re
for l in open('gpurun_out/r4_19_cull2.txt'):
    m=re.match(r'(\w+) +(\w+) +(\{.*?\})( identical=(\w+))?',l)
    if m:
        d=json.loads(m.group(3)); print(m.group(1), m.group(2), round(d.get('ms',0),3), d.get('solves'), d.get('evals'), m.group(5), d.get('error','')[:200])
PY
