#!/bin/bash
# headline kernel (65 536 envs, K = 1) and its neighbours for a list of variant libraries: un-profiled stamped times
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { BSKGPU_LIB=$R/basilisk_env_amd/variants/$1.so python3 bench.py --no-cpu-baseline --no-extra --full-line $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-6s %-36s value %.4g kernel_us %8.2f med %8.2f wall_us %8.2f' % ('$1', '$2', d['value'], r['kernel_us_stamped'], r.get('median_us',0), r['wall_us_per_launch']))"; }
for rep in 1 2; do for v in "$@"; do line $v "--steps 20000 --warmup 2000"; done; done
for v in "$@"; do line $v "--envs 32768 --steps 20000 --warmup 2000"; done
for v in "$@"; do line $v "--envs 131072 --steps 10000 --warmup 1000"; done
