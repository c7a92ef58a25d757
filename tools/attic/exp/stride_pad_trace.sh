#!/bin/bash
# The K = 1 launch under the kernel tracer with the slab's padding off / on, alternating on ONE box (the trace's figure for a 7 us kernel
# moves by 15 % from box to box: only a same-box comparison says anything).  Usage: tools/exp/stride_pad_trace.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pad_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in a b c; do for pad in 0 32; do
  export BSKGPU_STRIDE_PAD=$pad
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pad${pad}${rep} -- python3 $R/bench.py --no-cpu-baseline --no-extra --full-line --steps 40000 --warmup 4000 > $O/kt_pad${pad}${rep}.log 2>&1
done; done
cd $R; python3 - $O <<'PY'
import glob, os, sys
sys.path.insert(0, "tools")
import kernel_trace_summary as k
for d in sorted(glob.glob(sys.argv[1] + "/kt_pad*")):
    if not os.path.isdir(d): continue
    fs = glob.glob(d + "/*/*_kernel_trace.csv")
    s, _ = k.summarise(fs[0])
    print(os.path.basename(d), {x: round(s[x], 3) for x in ("median_us", "trimmed_mean_us", "p10_us", "p90_us") if x in s}, s.get("dispatches"))
PY
