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
    ks = [k for k in ks if os.path.getsize(k) > 0]
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
# HBM traffic per launch for the bench workloads: FETCH_SIZE is doubled (gfx950 reports 1/2 of a
# coalesced streaming read: MI355X_MICROARCH.md §HBM; calibrated on this kernel, profiles/r01/README.md)
traffic = {}
for tag, envs in (("65k", 65536), ("131k", 131072), ("4m", 4194304)):
    f, w = out.get("fetch_" + tag), out.get("write_" + tag)
    if f and w:
        fk = f["counters_mean_per_dispatch"]["FETCH_SIZE"]
        wk = w["counters_mean_per_dispatch"]["WRITE_SIZE"]
        traffic[str(envs)] = {"fetch_size_kb": fk, "write_size_kb": wk, "traffic_bytes": (2.0 * fk + wk) * 1024.0,
                              "correction": "2 x FETCH_SIZE + WRITE_SIZE (KB x 1024)"}
out["traffic"] = traffic
print(json.dumps(out, indent=1))
