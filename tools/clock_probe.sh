#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE cycles / kernel duration, from one rocprofv3 run (--kernel-trace + one PMC)
# usage: tools/clock_probe.sh <lib variant or -> <config> <points> [env...]
V=$1; C=$2; P=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
[ "$V" = "-" ] || export SVSDF_LIB_VARIANT=$V
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk_${V}_$C
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk_${V}_$C -o p -- python $ROOT/tools/prof_eval.py $C $P 4 > /tmp/clk.log 2>&1
python - <<PY
import csv, glob, collections
kt = glob.glob('/tmp/clk_${V}_$C/**/*kernel_trace.csv', recursive=True)[0]
cc = glob.glob('/tmp/clk_${V}_$C/**/*counter_collection.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'].split('(')[0].replace('void svsdf::', ''), int(r['End_Timestamp']) - int(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE' or r['Dispatch_Id'] not in dur: continue
    k, d = dur[r['Dispatch_Id']]
    a = agg[k[:40]]; a[0] += float(r['Counter_Value']); a[1] += d; a[2] += 1
for k, (c, d, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"{k:42s} launches {n:4d}  total {d/1e6:9.3f} ms  GRBM_GUI_ACTIVE/ns = {c/d:7.3f}")
PY
