#!/usr/bin/env python3
"""Print per-dispatch PMC counter values (rocprofv3 counter_collection csv) for kernels matching a substring."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = rows.setdefault(k, {"name": r["Kernel_Name"][:60], "grid": r.get("Grid_Size", "")})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({c for e in rows.values() for c in e if c not in ("name", "grid")})
print("disp grid " + " ".join(names))
for k, e in rows.items():
    print(k, e["grid"], " ".join("%.4g" % e.get(c, 0) for c in names), e["name"][:40])
