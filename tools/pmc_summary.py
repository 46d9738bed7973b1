"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel (last dispatch of each), counter values."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "composite|project|radix|tile_emit"
by = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    if not re.search(pat, name): continue
    by.setdefault((name, r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = {}
for (name, d), c in by.items():
    last[name] = c
for name, c in last.items():
    print(name)
    for k, v in c.items():
        print(f"    {k:28s} {v:16.0f}")
