#!/bin/bash
# A/B kernel library variants on one device.  Usage: tools/ab.sh "<bench args>" ["<bench args>" ...]
# (default: the full scenario at K = 1800 and K = 1).  Variants: basilisk_env_amd/variants/*.so (make variants).
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -eq 0 ] && set -- "--scenario full --substeps 1800 --steps 4 --warmup 1" "--scenario full --steps 200 --warmup 20"
for args in "$@"; do
  for lib in $R/basilisk_env_amd/variants/*.so; do
    name=$(basename "$lib" .so)
    BSKGPU_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s %-52s kernel_us %10.2f med %10.2f ms/step %9.4f vgpr %d'%('$name', '$args', r['kernel_us'], r.get('median_us',0), d['ms_per_step'], r['vgprs']))"
  done
done
