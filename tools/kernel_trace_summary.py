#!/usr/bin/env python3
"""Steady-state summary of rocprofv3 --kernel-trace runs of bench.py (tools/round.sh) -> kernel_trace.json.

usage: tools/kernel_trace_summary.py GPURUN_OUT_DIR [PROFILES_DIR]

For every kt_<name>/ run directory: the per-dispatch CSV of the step kernel, in dispatch order; the dispatches that
start within RAMP_MS of the kernel's first dispatch are set aside (clock ramp: the first ~25 ms of a run read up to
20 % long), at most half of them; over the rest median, p10 / p90, mean and the mean of the middle 80 %.  The result
carries the fingerprint of the kernel sources it was measured on (bench.kernel_fingerprint), and bench.py prices its
rooflines on max(its own stamped pass, this trimmed mean) only while the fingerprints agree.  With PROFILES_DIR the
per-dispatch durations are written next to the summary as kt_<name>_dispatches.csv (dispatch index, start offset
[us], duration [ns]) so that every figure can be recomputed from the committed files.
"""
import csv
import glob
import importlib.util
import json
import os
import re
import sys

RAMP_MS = 25.0
KEYS = {"kt_65k": "65k_k1", "kt_4m": "4m_k1", "kt_131k": "131k_k1", "kt_k1800": "bare_k1800", "kt_power_k1800": "power_k1800",
        "kt_full_k1800": "full_k1800", "kt_sh": "sh70"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fingerprint():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kernel_fingerprint()


def pct(srt, q):
    return srt[min(len(srt) - 1, max(0, int(round(q * (len(srt) - 1)))))]


def summarise(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "step_kernel" not in r.get("Kernel_Name", ""):
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("VGPR_Count"),
                         r.get("Accum_VGPR_Count"), r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")))
    if not rows:
        return None, []
    rows.sort()
    # one run may hold launches of different shapes (the stamped pass and warm-up are the same launch): keep the most
    # frequent (kernel, grid) pair
    shapes = {}
    for r in rows:
        shapes[(r[2], r[5])] = shapes.get((r[2], r[5]), 0) + 1
    top = max(shapes, key=shapes.get)
    rows = [r for r in rows if (r[2], r[5]) == top]
    t0 = rows[0][0]
    dur = [(r[0] - t0, r[1] - r[0]) for r in rows]
    n_ramp = sum(1 for s, _ in dur if s < RAMP_MS * 1e6)
    n_ramp = min(n_ramp, len(dur) // 2)
    steady = sorted(d for _, d in dur[n_ramp:])
    # the gap between a dispatch's end and the next one's start stamp (information only - nothing is priced on it): where it is zero the
    # start stamp coincides with the previous kernel's end and the duration includes front-end work that overlaps the previous
    # kernel's tail in an un-profiled stream (profiles/r05/trace_modes.txt: the 7 us headline kernel reads ~6.0 us with a gap, ~7.2 without)
    srt = rows[n_ramp:]
    gaps = sorted(srt[i + 1][0] - srt[i][1] for i in range(len(srt) - 1))
    cut = len(steady) // 10
    core = steady[cut:len(steady) - cut] if len(steady) >= 10 else steady
    rec = {"kernel": rows[0][2], "grid": rows[0][5], "workgroup": rows[0][6], "vgprs": rows[0][3], "agprs": rows[0][4],
           "dispatches": len(dur), "ramp_dispatches_dropped": n_ramp, "ramp_ms": RAMP_MS, "steady_dispatches": len(steady),
           "mean_all_us": sum(d for _, d in dur) / len(dur) / 1e3,
           "mean_ramp_us": (sum(d for _, d in dur[:n_ramp]) / n_ramp / 1e3) if n_ramp else None,
           "mean_us": sum(steady) / len(steady) / 1e3, "median_us": pct(steady, 0.5) / 1e3, "p10_us": pct(steady, 0.1) / 1e3,
           "p90_us": pct(steady, 0.9) / 1e3, "min_us": steady[0] / 1e3, "max_us": steady[-1] / 1e3,
           "trimmed_mean_us": sum(core) / len(core) / 1e3, "averaging": "mean of the middle 80 % of the steady-state dispatches",
           "gap_to_next_start_median_us": (pct(gaps, 0.5) / 1e3) if gaps else None,
           "share_of_dispatches_with_zero_gap": (sum(1 for g in gaps if g <= 0) / len(gaps)) if gaps else None}
    return rec, dur


def main():
    d = sys.argv[1]
    outdir = sys.argv[2] if len(sys.argv) > 2 else None
    out = {"fingerprint": fingerprint(), "runs": {}}
    extra_boxes = {}
    for sub in sorted(os.listdir(d)):
        p = os.path.join(d, sub)
        if not os.path.isdir(p) or not sub.startswith("kt_") or sub.startswith("kt_stats_"):      # (kt_stats_*: stats_kernel, collect_evidence.sh)
            continue
        fs = [f for f in glob.glob(os.path.join(p, "*", "*_kernel_trace.csv")) if os.path.getsize(f) > 0]
        if not fs:
            continue
        fs.sort(key=os.path.getmtime)            # (a local gpurun_out/ keeps earlier passes' files: the newest trace counts)
        rec, dur = summarise(fs[-1])
        if rec is None:
            continue
        rec["csv"] = "%s_dispatches.csv" % sub
        rec["stats_csv"] = "%s_kernel_stats.csv" % sub
        m = re.match(r"(kt_.+)_b(\d+)$", sub)
        if m:      # the same command traced on ANOTHER box (tools/kt_box.sh): kept beside the round's own run of that key
            extra_boxes.setdefault(KEYS.get(m.group(1), m.group(1)[3:]), []).append(rec)
        else:
            out["runs"][KEYS.get(sub, sub[3:])] = rec
        if outdir:
            with open(os.path.join(outdir, rec["csv"]), "w") as f:
                f.write("dispatch,start_offset_us,duration_ns\n")
                for i, (s, dd) in enumerate(dur):
                    f.write("%d,%.3f,%d\n" % (i, s / 1e3, dd))
            st = sorted(glob.glob(os.path.join(p, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
            if st:
                with open(st[-1]) as src, open(os.path.join(outdir, rec["stats_csv"]), "w") as dst:
                    dst.write(src.read())
    # A microsecond-scale kernel's figure in the profiler's trace depends on the box (the headline kernel: 7.3 ... 7.9 us over
    # four boxes, against 6.9 stamped on every one of them): where a key was traced on several boxes the MEDIAN of their
    # trimmed means is what bench.py uses.
    for key, recs in extra_boxes.items():
        if key in out["runs"]:
            allb = [out["runs"][key]] + recs
            out["runs"][key]["boxes"] = [{k: r[k] for k in ("trimmed_mean_us", "median_us", "p10_us", "p90_us", "steady_dispatches", "gap_to_next_start_median_us", "share_of_dispatches_with_zero_gap", "csv", "stats_csv")} for r in allb]
            tm = sorted(r["trimmed_mean_us"] for r in allb)
            out["runs"][key]["trimmed_mean_us_median_of_boxes"] = tm[len(tm) // 2] if len(tm) % 2 else 0.5 * (tm[len(tm) // 2 - 1] + tm[len(tm) // 2])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
