#!/bin/bash
# Artefacts for the harmonics forms BASELINE.json configs[4] names and round 1 rejected: the tree of commit 28b481e
# (extracted to _w28/, built there) run with its own bench.py: form 1 scalar-load stream, form 2 two-wave column split of
# the scalar stream, form 3 the whole coefficient stream resident in LDS ("LDS-tiled coefficient table").
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_rejected
mkdir -p $O
cd $R/_w28
for f in 1 2 3; do
  BSKGPU_SH_FORM=$f python3 bench.py --gravity sh --steps 300 --warmup 100 --no-cpu-baseline > $O/sh_form${f}_bench.json 2>> $O/err.log && echo form$f ok
done
cd /tmp && export TMPDIR=/tmp
cd $R/_w28 && BSKGPU_SH_FORM=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sh_lds -- python3 bench.py --gravity sh --steps 300 --warmup 100 --no-cpu-baseline > $O/kt_sh_lds.json 2> $O/kt_sh_lds.log && echo kt ok
find $O -name "*kernel_stats.csv" | head
