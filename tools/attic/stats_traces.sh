#!/bin/bash
# The batch-scalars kernels alone under the kernel tracer (the part of tools/round.sh that depends on csrc/bsk_aux.hip only):
# a request after every K = 1 step at four batch sizes, two-launch form and bsk_set_step_stats form.  Usage: tools/stats_traces.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ns in 65536 131072 1048576 4194304; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_stats_$ns -- python3 $R/tools/exp/stats_trace.py $ns > $O/kt_stats_$ns.log 2>&1 && echo kt_stats_$ns ok
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_stats_fused_$ns -- python3 $R/tools/exp/stats_trace.py $ns 3000 fused > $O/kt_stats_fused_$ns.log 2>&1 && echo kt_stats_fused_$ns ok
done
