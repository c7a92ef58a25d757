#!/bin/bash
# three-wave / pair / single-wave form of one library over batch sizes (full scenario, K = 1800; stamped kernel time).  Usage: tools/tri_occ.sh LIB.so [sizes...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$1; shift
for n in ${*:-1024 8192 12288 16384 20480}; do
  for form in tri pair single; do
    case $form in single) export BSKGPU_PAIR=0 BSKGPU_TRI=0;; pair) export BSKGPU_PAIR=1 BSKGPU_TRI=0;; tri) export BSKGPU_PAIR=0 BSKGPU_TRI=1;; esac
    BSKGPU_LIB=$LIB python3 $R/bench.py --no-cpu-baseline --no-extra --full-line --scenario full --substeps 1800 --steps 6 --warmup 3 --envs $n 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-7s envs %6s kernel_us %10.2f  %s'%('$form', '$n', r['kernel_us_stamped'], r['kernel']))"
  done
done
