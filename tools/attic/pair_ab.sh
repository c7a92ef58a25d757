#!/bin/bash
# pair form vs single-wave form of the same library, same box.  Usage: tools/pair_ab.sh [lib]
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${1:-$R/basilisk_env_amd/libbskgpu.so}
for args in "--scenario full --substeps 1800 --steps 20 --warmup 10" "--scenario power --substeps 1800 --steps 20 --warmup 10" "--scenario full --substeps 1800 --steps 10 --warmup 5 --envs 131072" "--scenario full --substeps 1800 --steps 20 --warmup 10 --envs 8192"; do
  for mode in 0 1; do
    BSKGPU_PAIR=$mode BSKGPU_LIB=$LIB python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pair=%s %-70s kernel_us %10.2f med %10.2f ms/step %9.4f vgpr %d %s'%('$mode', '$args', r['kernel_us_stamped'], r.get('median_us',0), d['ms_per_step'], r['vgprs'], r['kernel']))"
  done
done
