#!/bin/bash
# The headline kernel's rocprofv3 trace (and the same command un-profiled) on ONE MORE box: a 7 us kernel's figure in the
# profiler's trace depends on the box, so the round keeps several and bench.py uses their median (tools/kernel_trace_summary.py).
# Usage (one gpurun call per box): tools/kt_box.sh TAG IDX      -> gpurun_out/TAG/kt_65k_bIDX/, ab_65k_bIDX_plain.json
# afterwards, in this container: python tools/kernel_trace_summary.py gpurun_out/TAG gpurun_out/TAG > gpurun_out/TAG/kernel_trace.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}; IDX=${2:-2}
O=$R/gpurun_out/$TAG; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line"
cd /tmp && export TMPDIR=/tmp
$B --steps 40000 --warmup 4000 > $O/ab_65k_b${IDX}_plain.json 2>/dev/null && echo plain ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_65k_b$IDX -- $B --steps 40000 --warmup 4000 > $O/kt_65k_b$IDX.log 2>&1 && echo kt_65k_b$IDX ok
