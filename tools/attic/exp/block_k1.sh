#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do for b in 64 128 256; do for n in 65536 131072; do
BSKGPU_BLOCK=$b python3 bench.py --no-cpu-baseline --no-extra --full-line --envs $n --steps 20000 --warmup 2000 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('block %3d envs %6d value %.4g kernel_us %6.2f med %6.2f wall_us %6.2f' % ($b, $n, d['value'], r['kernel_us_stamped'], r.get('median_us',0), r['wall_us_per_launch']))"
done; done; done
