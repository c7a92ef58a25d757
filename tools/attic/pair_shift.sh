#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${1:-$R/basilisk_env_amd/libbskgpu.so}
for sh in 31 0 1 2 3 7 8 9 10; do
  BSKGPU_PAIR_SHIFT=$sh BSKGPU_PAIR=1 BSKGPU_LIB=$LIB python3 $R/bench.py --no-cpu-baseline --no-extra --full-line --scenario full --substeps 1800 --steps 10 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('shift %2s kernel_us %10.2f'%('$sh', r['kernel_us_stamped']))"
done
