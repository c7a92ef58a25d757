#!/bin/bash
# Scenario-level kernel times (65 536 envs): power / full at K = 1 and K = 1800.  Usage: tools/scen.sh TAG ["extra bench.py flags"]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
EXTRA=${2:-}
O=$R/gpurun_out
cd $R
for sc in bare power full; do
  python bench.py --no-cpu-baseline --no-extra --full-line $EXTRA --scenario $sc --steps 200 --warmup 20 > $O/scen_${TAG}_${sc}_k1.json 2>> $O/scen_$TAG.err
  python bench.py --no-cpu-baseline --no-extra --full-line $EXTRA --scenario $sc --substeps 1800 --steps 4 --warmup 1 > $O/scen_${TAG}_${sc}_k1800.json 2>> $O/scen_$TAG.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/scen_${TAG}_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print("%-40s kernel_us %10.2f ms/step %9.4f vgprs %s frac %s"%(f.split("/")[-1], r["kernel_us"], d["ms_per_step"], r["vgprs"], r.get("frac")))
    except Exception as e: print(f, "ERR", e)
PY
