#!/bin/bash
# the observation / reward rows padded too (BSKGPU_OSTRIDE_PAD elements; the slab keeps its 32): what consumer-visible contiguity costs the K = 1 launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { BSKGPU_OSTRIDE_PAD=$1 python3 bench.py --no-cpu-baseline --no-extra --full-line --steps 20000 --warmup 2000 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('output rows + %-4s elements: wall_us %6.3f stamped %6.3f' % ('$1', r['wall_us_per_launch'], r['kernel_us_stamped']))"; }
for rep in 1 2 3; do for pad in 0 32 96 224; do line $pad; done; done
