#!/bin/bash
# Cost of each feature inside the full-scenario kernel (65 536 envs, K = 1800): bench.py --features subsets.
# Usage: tools/features.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
cd $R
for f in power power,sun power,drag power,desat power,sun,drag power,sun,drag,desat; do
  python3 bench.py --no-cpu-baseline --no-extra --full-line --scenario full --features $f --substeps 1800 --steps 6 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s kernel_us %10.2f med %10.2f vgpr %d'%('$f', r['kernel_us'], r.get('median_us',0), r['vgprs']))"
done | tee $R/gpurun_out/features_$TAG.txt
