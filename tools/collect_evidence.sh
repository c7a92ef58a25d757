#!/bin/bash
# Copy what tools/round.sh's phases merged back under gpurun_out/TAG into profiles/TAG (tracked).  Called by `tools/round.sh TAG collect`.
# Usage (this container, repo root): tools/collect_evidence.sh r03
TAG=${1:-r05}
S=gpurun_out/$TAG; D=profiles/$TAG
mkdir -p $D
cp $S/kernel_trace.json $S/summary_latest.json $D/ 2>/dev/null
# the raw per-dispatch files and the un-profiled twins of every traced command: one sub-directory, so that the round's top level stays readable
mkdir -p $D/traces
cp $S/kt_*_dispatches.csv $S/kt_*_kernel_stats.csv $D/traces/ 2>/dev/null
cp $S/ab_*_plain.json $D/traces/ 2>/dev/null
[ -f $S/latency.json ] && cp $S/latency.json $D/latency_box.json
# the device-resident loop under rocprofv3 --kernel-trace --memory-copy-trace (tools/exp/rl_nocopy.py): kernels per name, copies
if [ -d $S/memcopy_rl ]; then
  python3 - $S/memcopy_rl $S/memcopy_rl.log > $D/memcopy_rl.txt <<'PY'
import csv, glob, sys, collections, re
d, log = sys.argv[1], sys.argv[2]
kt = glob.glob(d + "/*/*_kernel_trace.csv"); mc = glob.glob(d + "/*/*memory_copy*.csv")
print("rocprofv3 --kernel-trace --memory-copy-trace -- python3 tools/exp/rl_nocopy.py   (65 536 envs, full scenario, K = 1, 200 steps after reset_tensors)")
print("memory-copy records: %d file(s)%s" % (len(mc), "" if mc else "  -> none: no hipMemcpy* reached the copy engines or the tracer in the whole run"))
for f in mc:
    rows = list(csv.DictReader(open(f))); print("  ", f.split("/")[-1], len(rows), "records", collections.Counter(r.get("Direction", r.get("Kind", "?")) for r in rows))
c = collections.Counter()
for f in kt:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]; n = re.sub(r"<.*", "", n); n = n[:60]
        c[n] += 1
print("kernels dispatched (name prefix: count):")
for n, k in c.most_common():
    print("  %6d  %s" % (k, n))
for line in open(log):
    if "library copies" in line:
        print(line.strip())
PY
fi
# stats_kernel + stats_join_kernel per batch size (tools/exp/stats_trace.py under the kernel tracer): per-dispatch durations + a summary
python3 - $S $D <<'PY'
import csv, glob, json, os, sys
s, d = sys.argv[1], sys.argv[2]
out = {}
for run in sorted(glob.glob(s + "/kt_stats_*")):
    if not os.path.isdir(run):
        continue
    n = os.path.basename(run)[len("kt_stats_"):]           # "65536", or "fused_65536": bsk_set_step_stats, the join kernel alone
    rows = {"stats_kernel": [], "stats_join_kernel": []}
    fs = sorted(glob.glob(run + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
    for r in csv.DictReader(open(fs[-1])) if fs else []:
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if name.startswith("stats_join"):                  # stats_join_kernel (1 024 threads) or stats_join1_kernel<MAX_WAVES> (one wave, <= 4 096 waves of rewards)
            rows.setdefault("_join_name", name)
            name = "stats_join_kernel"
        if name in rows and name != "_join_name":
            rows[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X") or r.get("Grid_Size")))
    join_name = rows.pop("_join_name", None)
    if not rows["stats_join_kernel"]:
        continue
    os.makedirs(os.path.join(d, "traces"), exist_ok=True)
    with open(os.path.join(d, "traces", "kt_stats_%s.csv" % n), "w") as g:
        g.write("kernel,dispatch,start_offset_us,duration_ns,grid\n")
        t0 = min(v[0][0] for v in rows.values() if v)
        for name, v in rows.items():
            v.sort()
            for k, (t, dur, grid) in enumerate(v):
                g.write("%s,%d,%.3f,%d,%s\n" % (name, k, (t - t0) / 1e3, dur, grid))
    rec = {}
    for name, v in rows.items():
        if not v:
            continue
        dur = sorted(r[1] for r in v[len(v) // 4:])          # the first quarter set aside (clock ramp)
        cut = len(dur) // 10
        core = dur[cut:len(dur) - cut] or dur
        rec[name] = {"dispatches": len(v), "median_us": dur[len(dur) // 2] / 1e3, "trimmed_mean_us": sum(core) / len(core) / 1e3, "min_us": dur[0] / 1e3, "grid": v[0][2]}
    if join_name and "stats_join_kernel" in rec:
        rec["stats_join_kernel"]["kernel"] = join_name
    rec["both_trimmed_mean_us"] = sum(rec[k]["trimmed_mean_us"] for k in ("stats_kernel", "stats_join_kernel") if k in rec)
    out[n] = rec
if out:
    json.dump(out, open(os.path.join(d, "kt_stats.json"), "w"), indent=1)
    print("stats kernels (level 1 + join, us):", {k: "%.2f + %.2f" % (v.get("stats_kernel", {}).get("trimmed_mean_us", 0.0), v["stats_join_kernel"]["trimmed_mean_us"]) for k, v in out.items()})
PY
[ -f $S/isa_mix.json ] && cp $S/isa_mix.json $D/
[ -f gpurun_out/isa_$TAG/isa_mix.json ] && cp gpurun_out/isa_$TAG/isa_mix.json $D/
cp $S/bench_*.json $D/ 2>/dev/null        # NAME.json = the compact line the driver reads, NAME.extra.json = the whole record (tools/round.sh: line)
cp $S/gputest.log $D/gputest_tail.txt 2>/dev/null && tail -n 3 $S/gputest.log > $D/gputest_tail.txt
[ -d gpurun_out/${TAG}_lines ] && cp gpurun_out/${TAG}_lines/bench_*.json $D/ 2>/dev/null
python3 - $D <<'PY'
import json, sys, os
d = sys.argv[1]
if not (os.path.exists(os.path.join(d, "kernel_trace.json")) and os.path.exists(os.path.join(d, "isa_mix.json"))):
    sys.exit(0)
kt = json.load(open(os.path.join(d, "kernel_trace.json")))
im = json.load(open(os.path.join(d, "isa_mix.json")))
print("kernel_trace fingerprint", kt.get("fingerprint"), "isa_mix", im.get("_meta", {}).get("fingerprint"))
for k, v in kt.get("runs", {}).items():
    if k != "_meta":
        print("  %-14s %s" % (k, {a: v[a] for a in v if a in ("avg_us", "trimmed_mean_us", "dispatches", "n")}))
PY
