#!/bin/bash
# LDS-scratch variant against the register kernel at several batch sizes.  Usage: tools/ldss.sh [BSKGPU_LIB]
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$1" ] && export BSKGPU_LIB=$1
cd $R
for args in "--envs 65536 --steps 2000 --warmup 100" "--envs 131072 --steps 1000 --warmup 50" "--envs 262144 --steps 500 --warmup 30" "--envs 524288 --steps 300 --warmup 20" "--envs 4194304 --steps 50 --warmup 5" "--envs 65536 --substeps 1800 --steps 4 --warmup 1" "--envs 262144 --substeps 200 --steps 4 --warmup 1"; do
  for fl in "" "--lds-scratch"; do
    python3 bench.py --no-cpu-baseline --no-extra --full-line $args $fl 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline_hbm', d['roofline']); print('%-14s %-48s kernel_us %10.2f med %10.2f wall_us/step %10.2f vgpr %d lds %s'%('$fl' or 'registers', '$args', r['kernel_us'], r.get('median_us',0), d['ms_per_step']*1e3, r['vgprs'], r.get('lds_bytes')))"
  done
done
