#!/bin/bash
# rocprofv3 passes on the harmonics bench (config 5): kernel trace, then issue counters in their own pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-sh}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gravity sh --steps 20 --warmup 2"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sh -- $B > $O/kt_sh.log 2>&1 && echo kt_sh ok
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_sh -- $B > $O/sq_sh.log 2>&1 && echo sq_sh ok
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/mem_sh -- $B > $O/mem_sh.log 2>&1 && echo mem_sh ok
python3 $R/tools/prof_summary.py $O > $O/summary.json; cat $O/summary.json
