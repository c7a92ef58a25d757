#!/bin/bash
# Issue-side counters for any bench.py configuration.  Usage: tools/pmc_any.sh TAG <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $*"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -- $B > $O/a.log 2>&1 && echo a ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $O/b -- $B > $O/b.log 2>&1 && echo b ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1 && echo kt ok
python3 - <<PY
import csv,glob,collections,statistics
for sub in "ab":
    fs=glob.glob("$O/%s/*/*counter_collection.csv"%sub)
    if not fs: continue
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "step_kernel" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items(): print(k, int(statistics.mean(v[len(v)//4:])), "n=%d"%len(v))
fs=glob.glob("$O/kt/*/*kernel_stats.csv")
if fs:
    for r in list(csv.DictReader(open(fs[0])))[:2]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
