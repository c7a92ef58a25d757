#!/bin/bash
# config 5 (degree-70 harmonics, two-wave form): kernel time over batch sizes for variant libraries built with a 2- / 3-waves-per-SIMD
# register budget (profiles/r05/rejected/sh_three_waves.txt).  usage (GPU box): tools/exp/sh_waves.sh sh_w2 sh_w3
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for n in 65536 98304 131072 196608; do for v in "$@"; do
  BSKGPU_SH_FORM=5 BSKGPU_LIB=$R/basilisk_env_amd/variants/$v.so python3 bench.py --no-cpu-baseline --no-extra --full-line --gravity sh --envs $n --steps 600 --warmup 300 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-6s envs %7d  kernel_us %8.2f (median %8.2f) wall_us %8.2f  ns per spacecraft %.3f  vgprs %s grid %s' % ('$v', $n, r['kernel_us_stamped'], r.get('median_us',0), r['wall_us_per_launch'], r['wall_us_per_launch']*1e3/$n, r['vgprs'], r['grid']))"
done; done
