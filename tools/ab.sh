#!/bin/bash
# A/B kernel variants in one process-sequence on one device: K=1800, K=1 and large N
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/basilisk_env_amd/variants/*.so; do
  for rep in 1 2; do
  BSKGPU_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extra --substeps 1800 --steps 5 --warmup 1 2>/dev/null | python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-28s K1800 kernel_ms %.3f'%(os.path.basename('$lib') or 'default', d['roofline']['kernel_us']/1e3))"
  done
  BSKGPU_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-28s K1    kernel_us %.2f vgpr %d'%(os.path.basename('$lib') or 'default', d['roofline']['kernel_us'], d['roofline']['vgprs']))"
  BSKGPU_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extra --envs 4194304 --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-28s 4M    kernel_us %.1f'%(os.path.basename('$lib') or 'default', d['roofline']['kernel_us']))"
done
