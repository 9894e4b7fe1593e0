"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (sum over dispatches of the LAST n evaluations)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void svsdf::', '').replace('svsdf::', '')
    a[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id'])); n[k] += 1
names = sorted({c for v in a.values() for c in v})
print("kernel".ljust(28), "launches", " ".join(c.rjust(22) for c in names))
for k in sorted(a, key=lambda k: -a[k].get(names[0], 0)):
    print(k.ljust(28), str(n[k]).rjust(8), " ".join(("%.4g" % a[k].get(c, 0)).rjust(22) for c in names))
