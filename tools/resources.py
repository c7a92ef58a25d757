#!/usr/bin/env python3
"""Kernel resource table from hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin or a file).
usage: make -C basilisk_env_amd/csrc resources 2>&1 | python tools/resources.py [filter]"""
import re, sys, subprocess

def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
flt = sys.argv[1] if len(sys.argv) > 1 else "step_kernel"
print("%-60s %5s %5s %5s %6s %6s %4s %6s" % ("kernel", "sgpr", "vgpr", "agpr", "sspill", "vspill", "occ", "lds"))
for r in rows:
    d = demangle(r["name"])
    if flt not in d:
        continue
    d = re.sub(r"^void bsk::|\(.*$", "", d)
    print("%-60s %5s %5s %5s %6s %6s %4s %6s" % (d, r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"),
                                             r.get("VGPRs Spill"), r.get("Occupancy"), r.get("LDS Size")))
