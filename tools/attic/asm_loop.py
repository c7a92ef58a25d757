#!/usr/bin/env python3
"""Basic blocks of the K-th innermost loop ("Inner Loop Header") of one kernel from hipcc -S output, with instruction classes.
usage: tools/asm_loop.py FILE.s SYMBOL_SUBSTRING [K=0] [min_block=8]
(full-scenario single-wave kernel, round 4: inner loop 0 = the run of ticks with drag, 1 = without, ...)"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 0
minb = int(sys.argv[4]) if len(sys.argv) > 4 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
hdrs = [i for i in range(start, end) if "Inner Loop Header" in lines[i]]
print("inner loops at lines", [h + 1 for h in hdrs])
a = hdrs[K] - 1
b = hdrs[K + 1] - 1 if K + 1 < len(hdrs) else end


def cls(op):
    if op.startswith("v_accvgpr"): return "acc"
    if op.startswith("v_mov"): return "mov"
    if re.match(r"v_(fma|fmac|mul|add)_f64", op): return "f64"
    if re.match(r"v_(rcp|rsq)", op): return "trans"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"): return "sel"
    if op.startswith("v_readlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_"): return "vo"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "s"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global") or op.startswith("scratch"): return "mem"
    return None


cur = ["hdr", a, {}, []]
blocks = []
for i in range(a, b):
    l = lines[i]
    s = l.split(";")[0].strip()
    if not s:
        continue
    if s.endswith(":"):
        blocks.append(cur)
        cur = [s[:-1], i, {}, []]
        continue
    if s.startswith("."):
        continue
    op = s.split()[0]
    c = cls(op)
    if c:
        cur[2][c] = cur[2].get(c, 0) + 1
    if c == "br":
        cur[3].append(s)
    if "row_newbcast:13" in l:
        cur[2]["FACET"] = cur[2].get("FACET", 0) + 1
blocks.append(cur)
tot = {}
for bl in blocks:
    n = sum(v for k, v in bl[2].items() if k != "FACET")
    if n >= minb:
        print(bl[1] + 1, bl[0], n, bl[2], " | ".join(x.split()[-1] for x in bl[3]))
