#!/bin/bash
# Round 4 kernel A/B on one box: variant libraries basilisk_env_amd/variants/*.so (fast builds: J2 + 3 / 4 wheels only),
# un-profiled stamped kernel times at K = 1800 (bare / power / full) and K = 1, then one SQ counter pass per variant
# at the full level (VALU instructions per tick and wave).
# usage: tools/r04_ab.sh TAG [variant names...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
VARS=${*:-$(cd $R/basilisk_env_amd/variants && ls *.so | sed 's/\.so$//')}
cd $R
line() {  # name args
  BSKGPU_LIB=$R/basilisk_env_amd/variants/$1.so python3 bench.py --no-cpu-baseline --no-extra --full-line $2 2>>$O/err.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s %-44s kernel_us %10.2f med %10.2f wall_us %10.2f vgpr %d' % ('$1', '$2', r['kernel_us_stamped'], r.get('median_us',0), r['wall_us_per_launch'], r['vgprs']))"
}
for rep in 1 2; do
for args in "--scenario full --substeps 1800 --steps 6 --warmup 2" "--scenario power --substeps 1800 --steps 6 --warmup 2" "--substeps 1800 --steps 8 --warmup 2"; do
  for v in $VARS; do line $v "$args"; done
done
done | tee $O/ab.txt
for v in $VARS; do line $v "--steps 4000 --warmup 200"; done | tee -a $O/ab.txt
for v in $VARS; do line $v "--scenario full --steps 2000 --warmup 200"; done | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  for sc in full bare; do
    A="--substeps 600 --steps 3 --warmup 1"; [ $sc = full ] && A="--scenario full $A"
    BSKGPU_LIB=$R/basilisk_env_amd/variants/$v.so timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/pmc_${v}_$sc -- python3 $R/bench.py --no-cpu-baseline --no-extra --full-line $A > $O/pmc_${v}_$sc.log 2>&1
    python3 - <<PY | tee -a $O/ab.txt
import csv, glob, collections
tot = collections.Counter(); n = 0
for f in glob.glob("$O/pmc_${v}_$sc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "step_kernel" not in row.get("Kernel_Name", ""): continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
if tot.get("SQ_WAVES"):
    w = tot["SQ_WAVES"]; launches = w / 1024.0; ticks = 600.0
    print("%-12s %-5s VALU/tick/wave %8.1f SALU %7.1f valu_active %.3f wait_any %.3f" % ("$v", "$sc", tot["SQ_INSTS_VALU"] / w / ticks, tot["SQ_INSTS_SALU"] / w / ticks,
          tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1) , tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1)))
else:
    print("$v $sc: no counters")
PY
  done
done
