#!/bin/bash
# Where does the slab's 256-B row padding pay?  K = 1 launches at 32 768 / 65 536 / 98 304 / 131 072 spacecraft with the state slab's
# field rows padded by 0 / 32 elements, alternating on ONE box, three rounds; wall time per launch of the un-stamped loop.
# Needs the tunables build (make -C basilisk_env_amd/csrc tunables): the product library does not read BSKGPU_STRIDE_PAD.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export BSKGPU_LIB=$R/basilisk_env_amd/variants/tunables.so
line() { BSKGPU_STRIDE_PAD=$1 python3 bench.py --no-cpu-baseline --no-extra --full-line --envs $2 --steps $3 --warmup 2000 2>/dev/null | tail -n 1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('envs %-8s pad %-3s wall_us %7.3f value %.4g' % ('$2', '$1', d['ms_per_step']*1e3, d['value']))"; }
for rep in 1 2 3; do
  for n in 32768 65536 98304 131072; do for pad in 0 32; do line $pad $n 20000; done; done
done
for rep in 1 2; do for pad in 0 32; do line $pad 4194304 200; done; done
