#!/bin/bash
# One more bench configuration traced with rocprofv3 (and run un-profiled beside it), for a key round.sh does not cover.
# Usage (GPU box): tools/kt_extra.sh TAG NAME "<bench args>"      -> gpurun_out/TAG/kt_NAME/, ab_NAME_plain.json
# afterwards, in this container: python tools/kernel_trace_summary.py gpurun_out/TAG gpurun_out/TAG > gpurun_out/TAG/kernel_trace.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; NAME=$2; ARGS=$3
O=$R/gpurun_out/$TAG; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line"
cd /tmp && export TMPDIR=/tmp
$B $ARGS > $O/ab_${NAME}_plain.json 2>/dev/null && echo plain ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$NAME -- $B $ARGS > $O/kt_$NAME.log 2>&1 && echo kt_$NAME ok
