#!/usr/bin/env python3
"""Code-object facts of the SHIPPED library, per kernel: registers, spills, scratch, LDS - from the gfx950 code objects' own
.amdhsa metadata (llvm-readelf --notes) - and, for the harmonics kernels, where their scratch instructions sit (inside / outside the
Pines loop bodies, from the disassembly).  What a reader otherwise has to extract by hand.

usage (this container): tools/code_objects.py [LIB.so] [--json]      default: basilisk_env_amd/libbskgpu.so, the bench's kernels
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the kernels the bench lines launch: template arguments <GRAV, NRW, DIAG, FEAT, SPLIT> (bsk_kernels.hip) / rollout <GRAV, NRW, DIAG, ACT>
BENCH = [("step_kernel<1, 4, true, 0, 1>", "configs[2] headline / K = 1 800 bare (J2, 4 wheels)"),
         ("step_kernel<1, 4, true, 1, 1>", "power level"), ("step_kernel<1, 4, true, 2, 1>", "full scenario (the drop-in env's kernel)"),
         ("step_kernel<1, 4, true, 2, 3>", "full scenario, three-wave form (<= 16 384 spacecraft)"),
         ("step_kernel<2, 4, true, 0, 4>", "configs[4] harmonics, form 4"), ("step_kernel<2, 4, true, 0, 5>", "configs[4] harmonics, two-wave form 5 (65 536)"),
         ("rollout_kernel<1, 4, true, false>", "bsk_step_n, constant action"), ("rollout_kernel<1, 4, true, true>", "bsk_step_n, per-step device actions")]


def code_objects(lib, tmp):
    """(shared with tools/dpp_hazard.py: one offload bundle per translation unit, each unbundled on its own)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dpp_hazard", os.path.join(ROOT, "tools", "dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.code_objects(lib, tmp)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernels(obj):
    """-> [{name (mangled), vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, wg_max}] from the AMDGPU metadata note."""
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", obj], check=True, capture_output=True, text=True).stdout
    recs, cur = [], None
    for line in txt.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line.replace("- .", "  ."))
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":                    # first key of a kernel record (keys are sorted)
            cur = {}
            recs.append(cur)
        if cur is not None and k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                     "group_segment_fixed_size", "max_flat_workgroup_size", "name", "symbol"):
            if k == "name" and "name" in cur:    # (argument names come earlier in their own records; the kernel's own .name follows .max_flat...)
                pass
            cur[k] = v
    return [r for r in recs if r.get("symbol", "").endswith(".kd")]


def scratch_in_loops(obj, symbol):
    """Scratch (private-segment) instructions of one kernel: total, and those inside innermost loop bodies (a backward branch's span
    that contains no other backward branch) - for the harmonics kernels those are the Pines walks' bodies."""
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + symbol, obj], check=True, capture_output=True, text=True).stdout
    ins = []
    for line in txt.split("\n"):
        if not line.startswith("\t"):
            continue
        body, _, tail = line.strip().partition("//")
        m = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        if not m:
            continue
        addr = int(m.group(1), 16)
        op = body.split()[0]
        tgt = None
        if op.startswith("s_cbranch") or op == "s_branch":
            mt = re.search(r"<\S+?\+0x([0-9a-f]+)>", tail)
            tgt = int(mt.group(1), 16) if mt else None
        ins.append((addr, op, tgt))
    if not ins:
        return None
    base = ins[0][0]
    back = [(t + base if t is not None and t + base <= a else None, a) for a, op, t in ins if t is not None]
    loops = sorted((lo, hi) for lo, hi in back if lo is not None)
    inner = [(lo, hi) for lo, hi in loops if not any(l2 >= lo and h2 <= hi and (l2, h2) != (lo, hi) for l2, h2 in loops)]
    scr = [a for a, op, _ in ins if op.startswith("scratch_") or op.startswith("buffer_") and "offen" in op]
    scr = [a for a, op, _ in ins if op.startswith("scratch_")]
    inside = [a for a in scr if any(lo <= a <= hi for lo, hi in inner)]
    return {"scratch_instructions": len(scr), "inside_innermost_loops": len(inside), "innermost_loops": len(inner), "instructions": len(ins)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "basilisk_env_amd", "libbskgpu.so")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        found = []
        for obj in code_objects(lib, tmp):
            for r in kernels(obj):
                found.append((obj, r))
        names = demangle([r["symbol"][:-3] for _, r in found])
        for want, what in BENCH:
            for obj, r in found:
                d = names[r["symbol"][:-3]]
                if ("bsk::" + want + "(") in d.replace("void ", ""):
                    row = {"kernel": want, "what": what, "vgpr": int(r["vgpr_count"]), "agpr": int(r["agpr_count"]), "sgpr": int(r["sgpr_count"]),
                           "vgpr_spill": int(r["vgpr_spill_count"]), "sgpr_spill": int(r["sgpr_spill_count"]), "scratch_bytes": int(r["private_segment_fixed_size"]),
                           "lds_bytes": int(r["group_segment_fixed_size"]), "max_workgroup": int(r["max_flat_workgroup_size"])}
                    if want.startswith("step_kernel<2"):
                        row["scratch"] = scratch_in_loops(obj, r["symbol"][:-3])
                    rows.append(row)
                    break
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
        return
    print("%-38s %5s %5s %5s %7s %7s %8s %7s  %s" % ("kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "scratch", "lds", "what"))
    for r in rows:
        print("%-38s %5d %5d %5d %7d %7d %7dB %6dB  %s" % (r["kernel"], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_bytes"], r["lds_bytes"], r["what"]))
    for r in rows:
        if r.get("scratch"):
            s = r["scratch"]
            print("%s: %d scratch instructions of %d, %d of them inside the %d innermost loop bodies" % (r["kernel"], s["scratch_instructions"], s["instructions"], s["inside_innermost_loops"], s["innermost_loops"]))


if __name__ == "__main__":
    main()
