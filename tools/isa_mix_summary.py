#!/usr/bin/env python3
"""Summarise tools/isa_mix.sh counter passes into isa_mix.json (committed under profiles/rNN/; read by bench.py).

Per key: wave-level instruction counts of the step kernel per RK4 sub-step — SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64
and SQ_INSTS_VALU / SQ_INSTS_SALU divided by (SQ_WAVES x sub-steps per launch) over the dispatches with the
pass's sub-step count (the stamped / warm-up dispatches of the same command are identical launches)."""
import csv, glob, json, os, re, statistics, sys

d = sys.argv[1]
SUB = {"bare": 600, "power": 600, "full": 600, "sh": 2}
out = {}
for key, sub in SUB.items():
    rec = {}
    for suffix in ("f", "t"):
        fs = glob.glob(os.path.join(d, "%s_%s" % (key, suffix), "*", "*counter_collection.csv"))
        if not fs:
            continue
        per = {}
        for r in csv.DictReader(open(fs[0])):
            if "step_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        rec.update({k: statistics.mean(v) for k, v in per.items()})
        rec["dispatches_" + suffix] = len(next(iter(per.values()))) if per else 0
    if "SQ_WAVES" not in rec or "SQ_INSTS_VALU_FMA_F64" not in rec:
        continue
    w = rec["SQ_WAVES"] * sub
    m = {"fma": rec["SQ_INSTS_VALU_FMA_F64"] / w, "mul": rec["SQ_INSTS_VALU_MUL_F64"] / w,
         "add": rec["SQ_INSTS_VALU_ADD_F64"] / w, "trans": rec["SQ_INSTS_VALU_TRANS_F64"] / w,
         "substeps_per_launch": sub, "waves": rec["SQ_WAVES"], "lanes_per_env": rec["SQ_WAVES"] * 64.0 / 65536.0}
    if "SQ_INSTS_VALU" in rec:
        m.update({"valu": rec["SQ_INSTS_VALU"] / w, "salu": rec["SQ_INSTS_SALU"] / w,
                  "valu_active_over_wave_cycles": rec["SQ_ACTIVE_INST_VALU"] / rec["SQ_WAVE_CYCLES"],
                  "wait_any_over_wave_cycles": rec["SQ_WAIT_ANY"] / rec["SQ_WAVE_CYCLES"],
                  "grbm_gui_active": rec.get("GRBM_GUI_ACTIVE")})
    out[key] = m
# fingerprint of the kernel sources the counts were taken on (bench.py flags a mix counted on another tree)
import importlib.util
_spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
_b = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_b)
out["_meta"] = {"fingerprint": _b.kernel_fingerprint()}
print(json.dumps(out, indent=1))
