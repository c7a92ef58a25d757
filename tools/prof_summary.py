#!/usr/bin/env python3
"""Summarise a tools_prof.sh output directory into one JSON (committed under profiles/)."""
import csv, glob, json, os, statistics, sys

d = sys.argv[1]
out = {}
for sub in sorted(os.listdir(d)):
    p = os.path.join(d, sub)
    if not os.path.isdir(p):
        continue
    ks = glob.glob(os.path.join(p, "*", "*_kernel_stats.csv"))
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        out[sub] = {"kernel_stats": [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")} for r in rows[:4]]}
    cc = glob.glob(os.path.join(p, "*", "*_counter_collection.csv"))
    if cc:
        rows = [r for r in csv.DictReader(open(cc[0])) if "step_kernel" in r["Kernel_Name"]]
        per = {}
        for r in rows:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        meta = rows[0] if rows else {}
        out[sub] = {"dispatches": len(rows) // max(1, len(per)),
                    "grid": meta.get("Grid_Size"), "wg": meta.get("Workgroup_Size"), "vgpr": meta.get("VGPR_Count"),
                    "sgpr": meta.get("SGPR_Count"),
                    "counters_mean_per_dispatch": {k: statistics.mean(v[len(v) // 4:]) for k, v in per.items()}}
print(json.dumps(out, indent=1))
