#!/bin/bash
# The round's bench lines, run AFTER profiles/rNN/ holds this tree's kernel_trace.json / isa_mix.json / summary_latest.json,
# so that every line carries the committed rocprofv3 figures beside its own stamps.  Usage: tools/bench_lines.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/${TAG}_lines
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench.err && echo default ok
python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2>> $O/bench.err && echo steps20 ok
for sc in power full; do
  python bench.py --no-extra --scenario $sc --steps 2000 --warmup 200 > $O/bench_scenario_${sc}_k1.json 2>> $O/bench.err
  python bench.py --no-extra --scenario $sc --substeps 1800 --steps 20 --warmup 10 > $O/bench_scenario_${sc}_k1800.json 2>> $O/bench.err
done
python bench.py --no-extra --substeps 1800 --steps 20 --warmup 10 > $O/bench_bare_k1800.json 2>> $O/bench.err
python bench.py --gravity sh --steps 1000 --warmup 300 > $O/bench_sh.json 2>> $O/bench.err && echo sh ok
python bench.py --no-cpu-baseline --no-extra --full-line --envs 4194304 --steps 20 --warmup 3 > $O/bench_4m.json 2>> $O/bench.err
BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 200 --warmup 20 > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err && echo rehearsal ok
tail -3 $O/bench.err
