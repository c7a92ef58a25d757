#!/bin/bash
# Executed-instruction mix of the step kernel per RK4 sub-step and wave, for bench.py's fp64 rooflines.
# Usage (GPU box, repo root): tools/isa_mix.sh TAG [keys...]   (keys: bare power full sh; default all)
# Two separate --pmc passes per key (fp64 classes; totals), never combined with tracing.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}; shift
KEYS=${*:-bare power full sh}
O=$R/gpurun_out/isa_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for key in $KEYS; do
  case $key in
    bare)  ARGS="--substeps 600 --steps 3 --warmup 1" ;;
    power) ARGS="--scenario power --substeps 600 --steps 3 --warmup 1" ;;
    full)  ARGS="--scenario full --substeps 600 --steps 3 --warmup 1" ;;
    sh)    ARGS="--gravity sh --substeps 2 --steps 3 --warmup 1" ;;
  esac
  B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $ARGS"
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES --output-format csv -d $O/${key}_f -- $B > $O/${key}_f.log 2>&1 && echo ${key}_f ok
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/${key}_t -- $B > $O/${key}_t.log 2>&1 && echo ${key}_t ok
done
python3 $R/tools/isa_mix_summary.py $O > $O/isa_mix.json && cat $O/isa_mix.json
