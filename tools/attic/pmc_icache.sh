#!/bin/bash
# Instruction-cache counters of the step kernel for any bench.py configuration.  Usage: tools/pmc_icache.sh TAG <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift
O=$R/gpurun_out/pmci_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $*"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $O/a -- $B > $O/a.log 2>&1
python3 - <<PY
import csv,glob,collections,statistics
fs=glob.glob("$O/a/*/*counter_collection.csv")
per=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "step_kernel" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$TAG", {k:int(statistics.mean(v)) for k,v in per.items()})
PY
