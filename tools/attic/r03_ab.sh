#!/bin/bash
# A/B of the SAME bench command with and without rocprofv3 attached: the stamped pass of each run (bench.py's own
# dispatch stamps) and, for the profiled run, the profiler's per-dispatch trace.  Usage: tools/r03_ab.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ab() {  # name, bench args...
  name=$1; shift
  python3 $R/bench.py --no-cpu-baseline --no-extra --full-line "$@" > $O/ab_${name}_plain.json 2>> $O/ab.err && echo ab_${name}_plain ok
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 $R/bench.py --no-cpu-baseline --no-extra --full-line "$@" > $O/ab_${name}_rocprof.json 2> $O/kt_$name.log && echo ab_${name}_rocprof ok
}
ab 65k --steps 40000 --warmup 4000
ab full_k1800 --scenario full --substeps 1800 --steps 20 --warmup 10
ab sh --gravity sh --steps 1000 --warmup 300
cd $R
python3 tools/kernel_trace_summary.py $O $O > $O/kernel_trace.json && echo summary ok
