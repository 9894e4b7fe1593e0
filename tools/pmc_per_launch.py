"""Per-LAUNCH counters of one evaluation: python tools/pmc_per_launch.py <counter_collection.csv> [evaluation index]
Lists, in dispatch order, every kernel launch of the chosen evaluation (between two k_prep launches) with its counters --
e.g. SQ_WAVES / SQ_INSTS_VALU / SQ_INSTS_SALU per launch of the GSIP chain (which iteration pays how many instructions)."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = int(sys.argv[2]) if len(sys.argv) > 2 else 4
disp = collections.OrderedDict()
for r in rows:
    d = int(r["Dispatch_Id"])
    e = disp.setdefault(d, {"name": re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void svsdf::", "").replace("svsdf::", "")[:24], "c": {}, "grid": r.get("Grid_Size", "")})
    e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
preps = [i for i in ids if disp[i]["name"].startswith("k_prep")]
a, b = preps[ev], (preps[ev + 1] if ev + 1 < len(preps) else ids[-1] + 1)
names = sorted({c for i in ids for c in disp[i]["c"]})
print(f"{'launch':26s} " + " ".join(f"{n[:20]:>20s}" for n in names) + ("   VALU/wave  SALU/wave" if "SQ_WAVES" in names else ""))
for i in ids:
    if a <= i < b:
        c = disp[i]["c"]
        extra = ""
        if c.get("SQ_WAVES"):
            extra = f"   {c.get('SQ_INSTS_VALU', 0) / c['SQ_WAVES']:9.0f}  {c.get('SQ_INSTS_SALU', 0) / c['SQ_WAVES']:9.0f}"
        print(f"{disp[i]['name']:26s} " + " ".join(f"{c.get(n, 0.0):20.5g}" for n in names) + extra)
