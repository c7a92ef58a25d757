#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root).
# Each counter set gets its own pass; --pmc is never combined with tracing other than kernel-trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
B="python3 $R/bench.py --no-cpu-baseline --no-extra --full-line"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_65k -- $B > $O/kt_65k.log 2>&1 && echo kt_65k ok
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_65k -- $B --steps 50 > $O/fetch_65k.log 2>&1 && echo fetch_65k ok
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_65k -- $B --steps 50 > $O/write_65k.log 2>&1 && echo write_65k ok
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_65k -- $B --steps 50 > $O/sq_65k.log 2>&1 && echo sq_65k ok
BL="$B --envs 4194304 --steps 20 --warmup 3"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_4m -- $BL > $O/kt_4m.log 2>&1 && echo kt_4m ok
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_4m -- $BL > $O/fetch_4m.log 2>&1 && echo fetch_4m ok
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_4m -- $BL > $O/write_4m.log 2>&1 && echo write_4m ok
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_4m -- $BL > $O/sq_4m.log 2>&1 && echo sq_4m ok
BK="$B --substeps 1800 --steps 5 --warmup 1"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_k1800 -- $BK > $O/kt_k1800.log 2>&1 && echo kt_k1800 ok
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_k1800 -- $BK > $O/sq_k1800.log 2>&1 && echo sq_k1800 ok
find $O -name "*.csv" | head -40
