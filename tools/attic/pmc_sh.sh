#!/bin/bash
# Extra issue-side counters for the harmonics kernel (two passes).  Usage: tools/pmc_sh.sh TAG [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gravity sh --steps 10 --warmup 2 $*"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/a -- $B > $O/a.log 2>&1 && echo a ok
timeout -k 10 200 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -- $B > $O/b.log 2>&1 && echo b ok
python3 - <<PY
import csv,glob,collections,statistics
for sub in "ab":
    fs=glob.glob("$O/%s/*/*counter_collection.csv"%sub)
    if not fs: continue
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "step_kernel" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items(): print(k, int(statistics.mean(v[len(v)//4:])))
PY
