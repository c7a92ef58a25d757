#!/bin/bash
# Copy what tools/round.sh, tools/isa_mix.sh and tools/bench_lines.sh merged back under gpurun_out/ into profiles/TAG (tracked).
# Usage (this container, repo root): tools/collect_evidence.sh r03
TAG=${1:-r04}
S=gpurun_out/$TAG; D=profiles/$TAG
mkdir -p $D
cp $S/kernel_trace.json $S/summary_latest.json $D/ 2>/dev/null
cp $S/kt_*_dispatches.csv $S/kt_*_kernel_stats.csv $D/ 2>/dev/null
cp $S/ab_*_plain.json $D/ 2>/dev/null
[ -f $S/latency.json ] && cp $S/latency.json $D/latency_box.json
[ -f gpurun_out/isa_$TAG/isa_mix.json ] && cp gpurun_out/isa_$TAG/isa_mix.json $D/
[ -d gpurun_out/${TAG}_lines ] && cp gpurun_out/${TAG}_lines/bench_*.json $D/ 2>/dev/null
python3 - $D <<'PY'
import json, sys, os
d = sys.argv[1]
kt = json.load(open(os.path.join(d, "kernel_trace.json")))
im = json.load(open(os.path.join(d, "isa_mix.json")))
print("kernel_trace fingerprint", kt.get("fingerprint"), "isa_mix", im.get("_meta", {}).get("fingerprint"))
for k, v in kt.get("runs", {}).items():
    if k != "_meta":
        print("  %-14s %s" % (k, {a: v[a] for a in v if a in ("avg_us", "trimmed_mean_us", "dispatches", "n")}))
PY
