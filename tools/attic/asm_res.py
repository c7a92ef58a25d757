#!/usr/bin/env python3
"""Per-kernel resource table from hipcc -S output (the '; NumVgprs' comment blocks behind every function).
usage: tools/asm_res.py FILE.s [SYMBOL_SUBSTRING]"""
import re, sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "step_kernel"
name, rows, cur = None, [], {}
for l in open(path):
    m = re.match(r"\s+\.size\s+(\S+),", l)
    if m:
        name = m.group(1)
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.match(r"; (codeLenInByte|NumSgprs|NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize)\s*[:=]\s*(\d+)", l)
    if m and cur is not None:
        cur.setdefault(m.group(1), m.group(2))
print("%-34s %7s %5s %5s %5s %7s %4s" % ("kernel <GRAV,NRW,DIAG,FEAT,SPLIT>", "bytes", "sgpr", "vgpr", "agpr", "scratch", "occ"))
for r in rows:
    if flt not in r["name"]:
        continue
    m = re.search(r"ILi(\d)ELi(\d)ELb(\d)ELi(n?\d)ELi(\d)E", r["name"])
    short = "<%s>" % ",".join(x.replace("n", "-") for x in m.groups()) if m else r["name"][:34]
    print("%-34s %7s %5s %5s %5s %7s %4s" % (short, r.get("codeLenInByte"), r.get("NumSgprs"), r.get("NumVgprs"), r.get("NumAgprs"),
                                          r.get("ScratchSize"), r.get("Occupancy")))
