#!/bin/bash
# Executed VALU / SALU / LDS instructions and busy cycles per launch of the three forms at a small batch (GPU box).
# Usage: tools/tri_pmc.sh LIB N K
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${1:-$R/basilisk_env_amd/libbskgpu.so}; N=${2:-64}; K=${3:-1800}
O=$R/gpurun_out/tri_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for form in single pair tri; do
  case $form in single) export BSKGPU_PAIR=0 BSKGPU_TRI=0;; pair) export BSKGPU_PAIR=1 BSKGPU_TRI=0;; tri) export BSKGPU_PAIR=0 BSKGPU_TRI=1;; esac
  BSKGPU_LIB=$LIB timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/$form -- python3 $R/tools/exp/step_once.py $N $K 3 > $O/$form.log 2>&1
  python3 - $O/$form $form $K <<'PY'
import sys, glob, csv, collections
d, form, K = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "step_kernel" in r["Kernel_Name"]]
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: v[-1] for k, v in acc.items()}   # the last launch
w = out.get("SQ_WAVES", 1.0)
print(form, "waves %d" % w, " ".join("%s/tick/wg %.1f" % (k.replace("SQ_", ""), v / K / (w / {"single": 1, "pair": 2, "tri": 3}[form])) for k, v in sorted(out.items()) if k != "SQ_WAVES"))
PY
done
