#!/bin/bash
# THE evidence pass of a round - the single entry point.  One gpurun call per group of phases (a call is at most 20 minutes):
#
#   gpurun --timeout 1200 -- 'tools/round.sh r06 tests traces'      GPU suite; rocprofv3 kernel traces + un-profiled A/B lines
#   gpurun --timeout 1200 -- 'BOX=2 tools/round.sh r06 traces'      (and BOX=3) the kernel traces again on another box
#   gpurun --timeout 1200 -- 'tools/round.sh r06 counters isa'      HBM traffic / issue counter passes; executed instruction mix
#   gpurun --timeout 1200 -- 'tools/round.sh r06 lines'             the bench lines (after `collect` has put this tree's summaries
#                                                                   under profiles/TAG, so that every line prices on them)
#   tools/round.sh r06 collect        (this container, after each call)  gpurun_out/TAG -> profiles/TAG, README.md regenerated
#
# Outputs go to gpurun_out/TAG/ (scratch, merged back by gpurun).  Kernel traces and every --pmc set are separate runs; the
# program follows `--` directly.  Scripts this file does not call live in tools/attic/ (earlier rounds' experiments).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?usage: tools/round.sh TAG PHASE...}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extra --no-join --full-line"
prof() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1 ) && echo $tag ok || echo $tag FAILED; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
# name : bench arguments   (the keys of kernel_trace.json: tools/kernel_trace_summary.py)
RUNS=("65k:--steps 40000 --warmup 4000" "131k:--envs 131072 --steps 20000 --warmup 2000" "4m:--envs 4194304 --steps 20 --warmup 3"
      "k1800:--substeps 1800 --steps 20 --warmup 10" "power_k1800:--scenario power --substeps 1800 --steps 20 --warmup 10"
      "full_k1800:--scenario full --substeps 1800 --steps 20 --warmup 10" "sh:--gravity sh --steps 1000 --warmup 300")

phase_tests() {
  cd $R
  python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -2 $O/gputest.log
  python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -2 $O/smoke.log
}

phase_traces() {
  # BOX=2|3 in the environment: the same traces on ONE MORE box (a fresh gpurun call is a fresh box; the boxes of this pool differ by
  # up to 7 % on the fp64-bound kernels and a 7 us kernel's figure in the tracer depends on the box too): kt_<name>_b<BOX>, summarised
  # as the MEDIAN over the boxes (tools/kernel_trace_summary.py).  The stats / copy-tracer runs belong to box 1 only.
  sfx=${BOX:+_b$BOX}
  for run in "${RUNS[@]}"; do
    name=${run%%:*}; args=${run#*:}
    prof kt_$name$sfx --kernel-trace --stats --output-format csv -d $O/kt_$name$sfx -- $B $args
    ( cd $R && $B $args > $O/ab_${name}${sfx}_plain.json 2>> $O/bench.err )     # the same command un-profiled, same box
  done
  if [ -n "$BOX" ]; then cd $R && python3 tools/kernel_trace_summary.py $O $O > $O/kernel_trace.json && echo kernel_trace ok; return 0; fi
  phase_stats
  cd $R && python3 tools/kernel_trace_summary.py $O $O > $O/kernel_trace.json && echo kernel_trace ok
}

phase_stats() {   # (a phase of its own too: these depend on csrc/bsk_aux.hip only, which is not part of the step kernels' fingerprint)
  # the batch scalars (row a7): stats_kernel / stats_join*_kernel at four batch sizes, a request after every step
  for ns in 65536 131072 1048576 4194304; do
    prof kt_stats_$ns --kernel-trace --stats --output-format csv -d $O/kt_stats_$ns -- python3 $R/tools/exp/stats_trace.py $ns
    prof kt_stats_fused_$ns --kernel-trace --stats --output-format csv -d $O/kt_stats_fused_$ns -- python3 $R/tools/exp/stats_trace.py $ns 3000 fused
  done
  # the device-resident loop under the copy tracer: reset_tensors + step_tensors must show NO memory copy (row f4)
  prof memcopy_rl --kernel-trace --memory-copy-trace --output-format csv -d $O/memcopy_rl -- python3 $R/tools/exp/rl_nocopy.py
}

phase_counters() {
  for t in "65k:" "131k:--envs 131072" "4m:--envs 4194304 --steps 20 --warmup 3"; do
    name=${t%%:*}; args=${t#*:}; [ "$name" = 4m ] || args="$args --steps 50"
    prof fetch_$name --pmc FETCH_SIZE --output-format csv -d $O/fetch_$name -- $B $args
    prof write_$name --pmc WRITE_SIZE --output-format csv -d $O/write_$name -- $B $args
  done
  prof sq_65k --pmc $SQ --output-format csv -d $O/sq_65k -- $B --steps 50
  prof sq_4m --pmc $SQ --output-format csv -d $O/sq_4m -- $B --envs 4194304 --steps 20 --warmup 3
  prof sq_k1800 --pmc $SQ --output-format csv -d $O/sq_k1800 -- $B --substeps 1800 --steps 20 --warmup 10
  prof sq_full_k1800 --pmc $SQ --output-format csv -d $O/sq_full_k1800 -- $B --scenario full --substeps 1800 --steps 20 --warmup 10
  prof sq_sh --pmc $SQ --output-format csv -d $O/sq_sh -- $B --gravity sh --steps 1000 --warmup 300
  cd $R && python3 tools/prof_summary.py $O > $O/summary_latest.json 2>/dev/null && echo summary ok
}

phase_isa() {    # executed instruction mix per RK4 sub-step and wave, for bench.py's fp64 rooflines: two --pmc passes per key
  I=$O/isa; mkdir -p $I
  for key in bare power full sh; do
    case $key in
      bare)  ARGS="--substeps 600 --steps 3 --warmup 1" ;;
      power) ARGS="--scenario power --substeps 600 --steps 3 --warmup 1" ;;
      full)  ARGS="--scenario full --substeps 600 --steps 3 --warmup 1" ;;
      sh)    ARGS="--gravity sh --substeps 2 --steps 3 --warmup 1" ;;
    esac
    O=$I prof ${key}_f --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES --output-format csv -d $I/${key}_f -- $B $ARGS
    O=$I prof ${key}_t --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $I/${key}_t -- $B $ARGS
  done
  cd $R && python3 tools/isa_mix_summary.py $I > $O/isa_mix.json && echo isa_mix ok
}

# one bench command as the driver runs it: stdout = the EXTRA line + (LAST) the compact line -> NAME.json (the line the driver
# reads; refused above 4 096 bytes) and NAME.extra.json (the whole record)
line() {
  name=$1; shift
  ( cd $R && python bench.py "$@" > $O/$name.out 2>> $O/bench.err ) || { echo "$name FAILED (rc $?)"; return 1; }
  tail -n 1 $O/$name.out > $O/$name.json
  grep '^EXTRA ' $O/$name.out | tail -n 1 | sed 's/^EXTRA //' > $O/$name.extra.json
  sz=$(wc -c < $O/$name.json)
  if [ "$sz" -ge 4096 ]; then echo "$name: headline of $sz bytes exceeds the 4 096-byte limit"; return 1; fi
  echo "$name ok ($sz bytes)"
}

phase_lines() {
  rc=0
  line bench_default || rc=1
  line bench_steps20_warmup5 --steps 20 --warmup 5 || rc=1
  for sc in power full; do
    line bench_scenario_${sc}_k1 --no-extra --scenario $sc --steps 2000 --warmup 200 || rc=1
    line bench_scenario_${sc}_k1800 --no-extra --scenario $sc --substeps 1800 --steps 20 --warmup 10 || rc=1
  done
  line bench_bare_k1800 --no-extra --substeps 1800 --steps 20 --warmup 10 || rc=1
  line bench_sh --gravity sh --steps 1000 --warmup 300 || rc=1
  line bench_4m --no-cpu-baseline --no-extra --envs 4194304 --steps 20 --warmup 3 || rc=1
  BENCH_REHEARSAL=1 line bench_rehearsal4 --gpus 4 --steps 200 --warmup 20 || rc=1
  ( cd $R && python3 tools/latency.py > $O/latency.json 2>> $O/bench.err ) && echo latency ok
  tail -3 $O/bench.err
  return $rc
}

phase_collect() {   # this container: what the round keeps, tracked
  # (every box's traces are under gpurun_out/TAG by now: the summary over ALL of them is formed here, not on the last box)
  cd $R && ls -d $O/kt_65k* > /dev/null 2>&1 && python3 tools/kernel_trace_summary.py $O $O > $O/kernel_trace.json
  cd $R && bash tools/collect_evidence.sh $TAG && python3 tools/profiles_tables.py $TAG && echo "profiles/$TAG written"
}

rc=0
for ph in "$@"; do
  case $ph in
    tests|traces|stats|counters|isa|lines|collect) phase_$ph || rc=1 ;;
    *) echo "unknown phase $ph (tests traces stats counters isa lines collect)"; exit 2 ;;
  esac
done
exit $rc
