"""Sum rocprofv3 --pmc counters per kernel: python tools/pmc_agg.py <counter_collection.csv> [...]
Prints, per kernel name (template arguments kept, parameter list cut), dispatches and the sum of every counter."""
import csv
import collections
import re
import sys

for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    names = sorted({c for v in agg.values() for c in v})
    print("#", path)
    print(f"{'kernel':48s} {'disp':>6s} " + " ".join(f"{n[:22]:>22s}" for n in names))
    for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].values())):
        print(f"{k:48s} {len(disp[k]):6d} " + " ".join(f"{v.get(n, 0.0):22.4g}" for n in names))
