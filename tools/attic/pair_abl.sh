#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for args in "--scenario full --substeps 1800 --steps 10 --warmup 5 --envs 8192" "--scenario full --substeps 1800 --steps 10 --warmup 5"; do
for lib in $R/basilisk_env_amd/variants/*.so; do
  BSKGPU_PAIR=1 BSKGPU_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s %-60s kernel_us %10.2f'%('$(basename $lib .so)', '$args', r['kernel_us_stamped']))"
done; done
