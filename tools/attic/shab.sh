#!/bin/bash
# Harmonics kernel forms x library variants (65 536 envs, degree 70).  Usage: tools/shab.sh [envs]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-65536}
cd $R
for lib in $R/basilisk_env_amd/variants/*.so; do
  for form in 5; do
    BSKGPU_LIB=$lib BSKGPU_SH_FORM=$form python3 bench.py --no-cpu-baseline --no-extra --full-line --gravity sh --envs $N --steps 300 --warmup 150 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s form $form envs $N kernel_us %9.2f med %9.2f wall_us/step %9.2f vgpr %d  %s'%('$(basename $lib .so)', r['kernel_us'], r.get('median_us',0), d['ms_per_step']*1e3, r['vgprs'], r['kernel']))"
  done
done
