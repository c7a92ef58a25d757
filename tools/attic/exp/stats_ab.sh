#!/bin/bash
# stats_kernel under the kernel tracer at four batch sizes (+ its order tests), for the library BSKGPU_LIB names (default: in-tree)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
(cd $R && python -m pytest tests/test_gpu_device_surface.py -q -m gpu -k "batch_stats or snapshot or captured" 2>&1 | tail -1)
for ns in 65536 131072 1048576 4194304; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${1:-r05c}/kt_stats_$ns -- python3 $R/tools/exp/stats_trace.py $ns > $R/gpurun_out/${1:-r05c}_$ns.log 2>&1
  tail -1 $R/gpurun_out/${1:-r05c}_$ns.log
done
