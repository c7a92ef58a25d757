#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sq_ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in default exact; do
  lib=""; [ $v != default ] && lib=$R/basilisk_env_amd/variants/libbskgpu_$v.so
  export BSKGPU_LIB=$lib
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/$v -- python3 $R/bench.py --no-cpu-baseline --no-extra --substeps 1800 --steps 5 --warmup 1 > $O/$v.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_IFETCH --output-format csv -d $O/${v}_2 -- python3 $R/bench.py --no-cpu-baseline --no-extra --substeps 1800 --steps 5 --warmup 1 > $O/${v}_2.log 2>&1
done
python3 $R/tools/prof_summary.py $O
