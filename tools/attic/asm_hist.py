#!/usr/bin/env python3
"""Static opcode histogram of one kernel's loops from hipcc -S output.
usage: tools/asm_hist.py FILE.s SYMBOL_SUBSTRING [min_loop_len]
Splits the function at labels, reports every back-edge loop (label .. branch to it) with its instruction classes."""
import collections, re, sys

path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
insts = []
for l in body:
    s = l.split(";")[0].strip()
    if not s:
        continue
    if s.endswith(":"):
        labels[s[:-1]] = len(insts)
        continue
    if s.startswith("."):
        continue
    insts.append(s)


def cls(op):
    if op.startswith("v_accvgpr") : return "accvgpr"
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): return "fma64"
    if op.startswith("v_mul_f64"): return "mul64"
    if op.startswith("v_add_f64"): return "add64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op): return "trans"
    if op.startswith("v_mov") or op.startswith("v_pk_mov"): return "vmov"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp") : return "vcmp"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return op.split()[0]
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_") or op.startswith("buffer_"): return "scratch"
    if op.startswith("global_") or op.startswith("flat_"): return "global"
    return "other"

loops = []
for i, s in enumerate(insts):
    m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", s)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] <= i and i - labels[t] >= minlen:
            loops.append((labels[t], i, t))
print("function: %d instructions, %d loops >= %d" % (len(insts), len(loops), minlen))
for a, b, t in loops:
    h = collections.Counter(cls(s.split()[0]) for s in insts[a:b + 1])
    valu = sum(v for k, v in h.items() if k in ("fma64", "mul64", "add64", "trans", "vmov", "cndmask", "vcmp", "valu_other", "accvgpr", "lane"))
    print("loop %s: %d instr, VALU %d: %s" % (t, b - a + 1, valu, dict(h.most_common())))
    oth = collections.Counter(s.split()[0] for s in insts[a:b + 1] if cls(s.split()[0]) == "valu_other")
    print("   valu_other:", dict(oth.most_common(12)))
