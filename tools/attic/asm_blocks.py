#!/usr/bin/env python3
"""Basic blocks of one kernel from hipcc -S output, in layout order: size, fp64 arithmetic, moves, branch at the end.
usage: tools/asm_blocks.py FILE.s SYMBOL_SUBSTRING [min_block_len]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    s = l.split(";")[0].strip()
    cm = l.split(";")[1].strip() if ";" in l else ""
    if not s:
        if cm.startswith("%bb."):
            blocks.append((name, cur)); cur = []; name = cm.split()[0]
        continue
    if s.endswith(":"):
        blocks.append((name, cur)); cur = []; name = s[:-1] + ("  " + cm if cm else "")
        continue
    if s.startswith("."):
        continue
    cur.append(s)
blocks.append((name, cur))
pos = 0
for name, b in blocks:
    if len(b) >= minlen:
        h = collections.Counter()
        for s in b:
            op = s.split()[0]
            if op.startswith("v_fma") or op.startswith("v_mul_f64") or op.startswith("v_add_f64"): h["f64"] += 1
            elif op.startswith("v_accvgpr"): h["acc"] += 1
            elif op.startswith("v_mov") or op.startswith("v_pk_mov"): h["mov"] += 1
            elif op.startswith("v_cndmask") or op.startswith("v_cmp"): h["sel"] += 1
            elif re.match(r"v_(rcp|rsq|sqrt|exp|log)", op): h["trans"] += 1
            elif op.startswith("v_"): h["vother"] += 1
            elif op.startswith("scratch_"): h["scratch"] += 1
            elif op.startswith("ds_"): h["lds"] += 1
            elif op.startswith("global_"): h["glob"] += 1
            elif op.startswith("s_waitcnt"): h["wait"] += 1
            elif op.startswith("s_nop"): h["nop"] += 1
            elif op.startswith("s_"): h["s"] += 1
        br = [s for s in b if s.startswith("s_cbranch") or s.startswith("s_branch")]
        print("%6d %-44s n=%4d %s  %s" % (pos, name[:44], len(b), dict(h), " ".join(x.split()[-1] for x in br[-2:])))
    pos += len(b)
