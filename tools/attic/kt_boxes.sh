#!/bin/bash
# Every bench configuration's rocprofv3 kernel trace (and the same command un-profiled) on ONE MORE box.  The boxes of this pool
# differ by up to 7 % on the fp64-bound kernels (the clock they hold under a dense fp64 load) and a 7 us kernel's figure in the
# profiler's trace depends on the box too, so the round keeps three boxes per key and bench.py prices its rooflines on the MEDIAN
# of their steady-state trimmed means (tools/kernel_trace_summary.py: kt_<name>_b<IDX>).
# Usage (one gpurun call per box): tools/kt_boxes.sh TAG IDX
# afterwards, in this container: python tools/kernel_trace_summary.py gpurun_out/TAG gpurun_out/TAG > gpurun_out/TAG/kernel_trace.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}; IDX=${2:-2}
O=$R/gpurun_out/$TAG; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line"
cd /tmp && export TMPDIR=/tmp
run() {  # name args
  $B $2 > $O/ab_$1_b${IDX}_plain.json 2>/dev/null
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1_b$IDX -- $B $2 > $O/kt_$1_b$IDX.log 2>&1 && echo kt_$1_b$IDX ok
}
run 65k "--steps 40000 --warmup 4000"
run 131k "--envs 131072 --steps 20000 --warmup 2000"
run 4m "--envs 4194304 --steps 20 --warmup 3"
run k1800 "--substeps 1800 --steps 20 --warmup 10"
run power_k1800 "--scenario power --substeps 1800 --steps 20 --warmup 10"
run full_k1800 "--scenario full --substeps 1800 --steps 20 --warmup 10"
run sh "--gravity sh --steps 1000 --warmup 300"
