#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() { BSKGPU_LIB=$R/basilisk_env_amd/variants/$1.so python3 bench.py --no-cpu-baseline --no-extra --full-line $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-6s %-40s kernel_us %9.2f med %9.2f wall_us %9.2f vgpr %d' % ('$1', '$2', r['kernel_us_stamped'], r.get('median_us',0), r['wall_us_per_launch'], r['vgprs']))"; }
for rep in 1 2 3; do for v in base p20; do line $v "--envs 4194304 --steps 30 --warmup 5"; done; done
for rep in 1 2; do for v in base p20; do line $v "--envs 131072 --steps 20000 --warmup 2000"; done; done
for rep in 1 2; do for v in base p20; do line $v "--steps 40000 --warmup 4000"; done; done
