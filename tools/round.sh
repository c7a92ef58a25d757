#!/bin/bash
# One GPU-box pass of the round's evidence: GPU tests, bench lines, the 2-rank rehearsal of the self-launch path on
# one card, kernel traces and counter passes.  Usage: tools/round.sh TAG   (outputs under gpurun_out/TAG/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -2 $O/gputest.log
python bench.py > $O/bench_default.json 2> $O/bench.err && echo bench ok
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench.err && echo bench20 ok
BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 200 --warmup 20 > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err && echo rehearsal ok
for sc in power full; do
  python bench.py --no-cpu-baseline --no-extra --scenario $sc --steps 500 --warmup 50 > $O/bench_${sc}_k1.json 2>> $O/bench.err
  python bench.py --no-cpu-baseline --no-extra --scenario $sc --substeps 1800 --steps 20 --warmup 10 > $O/bench_${sc}_k1800.json 2>> $O/bench.err
done
python bench.py --no-cpu-baseline --gravity sh --steps 2000 --warmup 300 > $O/bench_sh.json 2>> $O/bench.err && echo sh ok
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra"
prof() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1 && echo $tag ok; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
prof kt_65k --kernel-trace --stats --output-format csv -d $O/kt_65k -- $B --steps 40000 --warmup 4000
prof fetch_65k --pmc FETCH_SIZE --output-format csv -d $O/fetch_65k -- $B --steps 50
prof write_65k --pmc WRITE_SIZE --output-format csv -d $O/write_65k -- $B --steps 50
prof sq_65k --pmc $SQ --output-format csv -d $O/sq_65k -- $B --steps 50
BL="$B --envs 4194304 --steps 20 --warmup 3"
prof kt_4m --kernel-trace --stats --output-format csv -d $O/kt_4m -- $BL
prof fetch_4m --pmc FETCH_SIZE --output-format csv -d $O/fetch_4m -- $BL
prof write_4m --pmc WRITE_SIZE --output-format csv -d $O/write_4m -- $BL
prof sq_4m --pmc $SQ --output-format csv -d $O/sq_4m -- $BL
B3="$B --envs 131072 --steps 20000 --warmup 2000"       # BASELINE configs[3], one GPU's share
prof kt_131k --kernel-trace --stats --output-format csv -d $O/kt_131k -- $B3
prof fetch_131k --pmc FETCH_SIZE --output-format csv -d $O/fetch_131k -- $B --envs 131072 --steps 50
prof write_131k --pmc WRITE_SIZE --output-format csv -d $O/write_131k -- $B --envs 131072 --steps 50
BK="$B --substeps 1800 --steps 20 --warmup 10"
prof kt_k1800 --kernel-trace --stats --output-format csv -d $O/kt_k1800 -- $BK
prof sq_k1800 --pmc $SQ --output-format csv -d $O/sq_k1800 -- $BK
BF="$B --scenario full --substeps 1800 --steps 20 --warmup 10"
prof kt_full_k1800 --kernel-trace --stats --output-format csv -d $O/kt_full_k1800 -- $BF
prof sq_full_k1800 --pmc $SQ --output-format csv -d $O/sq_full_k1800 -- $BF
BP="$B --scenario power --substeps 1800 --steps 20 --warmup 10"
prof kt_power_k1800 --kernel-trace --stats --output-format csv -d $O/kt_power_k1800 -- $BP
BS="$B --gravity sh --steps 1000 --warmup 300"
prof kt_sh --kernel-trace --stats --output-format csv -d $O/kt_sh -- $BS
prof sq_sh --pmc $SQ --output-format csv -d $O/sq_sh -- $BS
# the batch scalars on demand (row a7): stats_kernel's own duration at four batch sizes, a request after every step
for ns in 65536 131072 1048576 4194304; do
  prof kt_stats_$ns --kernel-trace --stats --output-format csv -d $O/kt_stats_$ns -- python3 $R/tools/exp/stats_trace.py $ns
  prof kt_stats_fused_$ns --kernel-trace --stats --output-format csv -d $O/kt_stats_fused_$ns -- python3 $R/tools/exp/stats_trace.py $ns 3000 fused
done
cd $R
python3 tools/prof_summary.py $O > $O/summary_latest.json 2>/dev/null && echo summary ok
python3 tools/kernel_trace_summary.py $O $O > $O/kernel_trace.json && echo kernel_trace ok
# A/B: the same commands without the profiler (their own stamped pass), beside the traces above
# the device-resident loop under the copy tracer: reset_tensors + step_tensors must show NO memory copy (row f4)
prof memcopy_rl --kernel-trace --memory-copy-trace --output-format csv -d $O/memcopy_rl -- python3 $R/tools/exp/rl_nocopy.py
for ab in "131k:--envs 131072 --steps 20000 --warmup 2000" "65k:--steps 40000 --warmup 4000" "full_k1800:--scenario full --substeps 1800 --steps 20 --warmup 10" "power_k1800:--scenario power --substeps 1800 --steps 20 --warmup 10" "k1800:--substeps 1800 --steps 20 --warmup 10" "sh:--gravity sh --steps 1000 --warmup 300" "4m:--envs 4194304 --steps 20 --warmup 3"; do
  python bench.py --no-cpu-baseline --no-extra ${ab#*:} > $O/ab_${ab%%:*}_plain.json 2>> $O/bench.err
done
python3 tools/latency.py > $O/latency.json 2>> $O/bench.err && echo latency ok
