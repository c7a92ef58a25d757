#!/bin/bash
# rollout kernel variants (basilisk_env_amd/variants/<name>.so): wall time per env step, then counter passes of the first one
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/rollout_ab; mkdir -p $O
for rep in 1 2; do for v in "$@"; do echo "== $v"; BSKGPU_LIB=$R/basilisk_env_amd/variants/$v.so python3 tools/exp/rollout_time.py 65536 1 2>/dev/null | grep -E "T 541|no history|name"; done; done
for v in "$@"; do echo "== $v 4Mi"; BSKGPU_LIB=$R/basilisk_env_amd/variants/$v.so python3 tools/exp/rollout_time.py 4194304 1 2>/dev/null | grep -E "T 100|name"; done
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
export BSKGPU_LIB=$R/basilisk_env_amd/variants/$v.so
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq_$v -- python3 $R/tools/exp/rollout_once.py > $O/sq_$v.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/ic_$v -- python3 $R/tools/exp/rollout_once.py > $O/ic_$v.log 2>&1
python3 - $O $v <<'PY'
import csv, glob, collections, sys
o, v = sys.argv[1], sys.argv[2]
for tag in ("sq", "ic"):
    per = collections.defaultdict(float)
    for f in glob.glob("%s/%s_%s/*/*counter_collection.csv" % (o, tag, v)):
        for r in csv.DictReader(open(f)):
            if "rollout_kernel" in r["Kernel_Name"]:
                per[r["Counter_Name"]] += float(r["Counter_Value"])
    print(v, tag, {k: int(x) for k, x in sorted(per.items())})
PY
done
