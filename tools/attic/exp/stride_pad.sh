#!/bin/bash
# K = 1 launches with the state slab's field rows padded by BSKGPU_STRIDE_PAD elements (the observation rows keep their stride): do ~46 rows
# at a power-of-two distance collide in the L2 channels?  (default build: 32 elements = 256 B)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { BSKGPU_STRIDE_PAD=$1 python3 bench.py --no-cpu-baseline --no-extra --full-line $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pad %-6s %-40s value %.4g stamped_us %8.2f wall_us %8.2f' % ('$1', '$2', d['value'], r['kernel_us_stamped'], r['wall_us_per_launch']))"; }
for rep in 1 2; do for pad in 0 32 64 96 160 512 544 4128; do line $pad "--steps 20000 --warmup 2000"; done; done
for pad in 0 32 96 544; do line $pad "--envs 131072 --steps 10000 --warmup 1000"; done
for pad in 0 32 96 544; do line $pad "--envs 4194304 --steps 30 --warmup 5"; done
for pad in 0 32; do line $pad "--substeps 1800 --steps 20 --warmup 10"; line $pad "--scenario full --substeps 1800 --steps 20 --warmup 10"; line $pad "--gravity sh --steps 1000 --warmup 300"; done
