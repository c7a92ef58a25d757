#!/bin/bash
# One GPU-box pass of the round's evidence: GPU tests, the default bench line, the 2-rank rehearsal of the
# self-launch path on one card, then the profile passes.  Usage: tools/round.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gputest_$TAG.log 2>&1; tail -3 $O/gputest_$TAG.log
python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err && echo bench ok
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_${TAG}_steps20.json 2>> $O/bench_$TAG.err && echo bench20 ok
BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 200 --warmup 20 > $O/bench_${TAG}_rehearsal2.json 2> $O/bench_${TAG}_rehearsal2.err && echo rehearsal ok
