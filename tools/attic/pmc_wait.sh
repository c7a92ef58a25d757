#!/bin/bash
# Where a kernel's waves wait: memory-side SQ counters for any bench.py configuration.  Usage: tools/pmc_wait.sh TAG <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift
O=$R/gpurun_out/pmcw_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $*"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/a -- $B > $O/a.log 2>&1 && echo a ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $O/b -- $B > $O/b.log 2>&1 && echo b ok
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/c -- $B > $O/c.log 2>&1 && echo c ok
python3 - <<PY
import csv,glob,collections,statistics
for sub in "abc":
    fs=glob.glob("$O/%s/*/*counter_collection.csv"%sub)
    if not fs: continue
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "step_kernel" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items(): print(k, int(statistics.mean(v)), "n=%d"%len(v))
PY
