#!/bin/bash
# A/B kernel library variants on one device: large-N K=1, headline K=1, K=1800
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/basilisk_env_amd/variants/*.so; do
  name=$(basename "$lib"); [ -z "$lib" ] && name=default
  for args in "--envs 4194304 --steps 20 --warmup 3" "--envs 1048576 --steps 40 --warmup 5" "--steps 200 --warmup 20" "--substeps 1800 --steps 5 --warmup 1"; do
    BSKGPU_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extra $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %-44s kernel_us %10.2f vgpr %d'%('$name', '$args', d['roofline']['kernel_us'], d['roofline']['vgprs']))"
  done
done
