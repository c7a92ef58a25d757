#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for LIB in "$@"; do
for n in 16384 49152 65536; do
  BSKGPU_PAIR=1 BSKGPU_LIB=$R/basilisk_env_amd/variants/$LIB.so python3 $R/bench.py --no-cpu-baseline --no-extra --full-line --scenario full --substeps 1800 --steps 6 --warmup 3 --envs $n 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-12s envs %6s kernel_us %10.2f'%('$LIB', '$n', r['kernel_us_stamped']))"
done; done
