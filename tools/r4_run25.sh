#!/bin/bash
# round 4, GPU call 25: anchor scans (Lipschitz skip of sample scans in the full-scan GSIP mode) vs the committed build
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
rm -f gpurun_out/r4_25_anchor.txt
for c in C3:1000000 C4:1000000 C5:300000 C3:100000 C3:1000000; do
  timeout 600 python tools/exp_variants.py an1 ${c%%:*} ${c##*:} >> gpurun_out/r4_25_anchor.txt 2>&1
done
python - <<'PY'
import json,re
for l in open('gpurun_out/r4_25_anchor.txt'):
    m=re.match(r'(\w+) +(\w+) +(\{.*?\})( identical=(\w+))?',l)
    if m:
        d=json.loads(m.group(3)); print(m.group(1), m.group(2), round(d.get('ms',0),3), d.get('solves'), d.get('evals'), d.get('scan'), m.group(5), d.get('error','')[:300])
PY
