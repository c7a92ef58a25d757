#!/usr/bin/env python3
"""DPP read-after-write hazards in the SHIPPED code objects.

usage: tools/dpp_hazard.py LIB.so [LIB.so ...]        (exit code 1 when a hazard is found)

The kernels feed wave-uniform constants to their FMAs through `v_fmac_f64_dpp ... row_newbcast:N` / `v_mov_b64_dpp`, written
as inline asm (bsk_device.hpp: fmac_k, get_k): there is no builtin for an fp64 DPP FMA.  gfx950 wants two wait states between
a VALU write of a VGPR and a DPP read of it (five after a VALU write of EXEC), the compiler's hazard recognizer pads them -
and does not look inside inline asm.  A table register the allocator parks in an AGPR and reloads right in front of its next
use is exactly that hazard (seen once: round 4, the density's rare full-evaluation path; the DPP read returned the stale
register).  So the check is made on what ships: every gfx950 code object in the library is disassembled and every DPP
instruction's broadcast source is traced back over its fall-through predecessors and over every branch that targets it.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DPP_VGPR_WAIT, DPP_EXEC_WAIT = 2, 5


def regs(tok):
    """'v[20:21]' / 'v7' -> set of VGPR numbers (None for anything else)"""
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return None


BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp):
    """Every gfx950 code object of the library -> list of ELF paths.  The library's .hip_fatbin section holds ONE offload bundle per
    translation unit, back to back (eleven since the step kernel was cut into eight units), and clang-offload-bundler reads only the
    first bundle of a file: the section is cut at every bundle magic and each piece unbundled on its own."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(BUNDLE_MAGIC), blob)]
    objs = []
    for k, lo in enumerate(starts):
        piece = os.path.join(tmp, "bundle%d.bin" % k)
        with open(piece, "wb") as f:
            f.write(blob[lo:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        out = subprocess.run([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + piece], check=True, capture_output=True, text=True).stdout
        for i, tgt in enumerate(t for t in out.split() if "gfx950" in t):
            obj = os.path.join(tmp, "code%d_%d.o" % (k, i))
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + tgt, "--input=" + piece, "--output=" + obj], check=True)
            if os.path.getsize(obj) > 0:
                objs.append(obj)
    return objs


def disassemble(lib, tmp):
    return [subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", obj], check=True, capture_output=True, text=True).stdout
            for obj in code_objects(lib, tmp)]


def parse(text):
    """-> {function: [(address, mnemonic, [operand tokens], branch target address or None)]}"""
    funcs, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body, _, tail = line.strip().partition("//")
        m = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        if not m:
            continue
        addr = int(m.group(1), 16)
        parts = body.strip().split(None, 1)
        op = parts[0]
        ops = [t.strip() for t in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        tgt = None
        if op.startswith("s_cbranch") or op == "s_branch":
            mt = re.search(r"<\S+?\+0x([0-9a-f]+)>", tail)
            mf = re.search(r"<(\S+?)>", tail)
            if mt:
                tgt = ("off", int(mt.group(1), 16))
            elif mf:
                tgt = ("off", 0)
        cur.append((addr, op, ops, tgt))
    return funcs


def check(funcs):
    bad, n_dpp = [], 0
    for name, ins in funcs.items():
        if not ins:
            continue
        base = ins[0][0]
        index = {a: i for i, (a, _, _, _) in enumerate(ins)}
        preds = {}
        for i, (a, op, ops, tgt) in enumerate(ins):
            if tgt is not None and base + tgt[1] in index:
                preds.setdefault(index[base + tgt[1]], []).append(i)

        def walk(i, need_v, need_e, src, seen):
            """instructions that execute right before ins[i]: fall-through and branches; -> offending instruction index or None"""
            stack = [(i, 0)]
            while stack:
                j, ws = stack.pop()
                if ws >= max(need_v, need_e):
                    continue
                for p in preds.get(j, []):           # arrived by a taken branch: the branch itself is one wait state
                    if (p, ws + 1) not in seen:
                        seen.add((p, ws + 1))
                        stack.append((p, ws + 1))
                if j == 0:
                    continue
                k = j - 1
                a, op, ops, tgt = ins[k]
                if op == "s_branch" or op in ("s_endpgm", "s_setpc_b64"):
                    continue                         # no fall-through from an unconditional transfer
                if op.startswith("v_") and ops:
                    w = regs(ops[0])
                    if (op.startswith("v_swap_") or "permlane16_swap" in op or "permlane32_swap" in op) and len(ops) > 1:
                        w = (w or set()) | (regs(ops[1]) or set())      # two-destination operations write both operands
                    if w and (w & src) and ws < need_v:
                        return k
                    if op.startswith("v_cmpx") and ws < need_e:
                        return k
                step = 1
                if op == "s_nop":
                    step = int(ops[0], 0) + 1
                if (k, ws + step) not in seen:
                    seen.add((k, ws + step))
                    stack.append((k, ws + step))
            return None

        for i, (a, op, ops, tgt) in enumerate(ins):
            if "_dpp" not in op:
                continue
            n_dpp += 1
            src = regs(ops[1].split()[0]) if len(ops) > 1 else None
            if not src:
                continue
            k = walk(i, DPP_VGPR_WAIT, DPP_EXEC_WAIT, src, set())
            if k is not None:
                bad.append((name, ins[k], ins[i]))
    return n_dpp, bad


TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_clamp|_legacy)?_f(16|32|64)")


def check_trans(funcs):
    """gfx940+ forwarding hazard: a VALU instruction must not read a transcendental's result in the very next issue slot
    (one wait state).  The compiler pads its own instructions; an inline-asm consumer it would not see.  Checked for EVERY
    VALU consumer (fall-through only: a branch in between is a wait state)."""
    bad = []
    for name, ins in funcs.items():
        for i in range(1, len(ins)):
            a0, op0, ops0, _ = ins[i - 1]
            a1, op1, ops1, _ = ins[i]
            if not TRANS.match(op0) or not op1.startswith("v_") or not ops0:
                continue
            w = regs(ops0[0])
            if not w:
                continue
            for t in ops1[1:]:
                r = regs(t.split()[0])
                if r and (r & w):
                    bad.append((name, ins[i - 1], ins[i]))
                    break
    return bad


def main():
    rc = 0
    for lib in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as tmp:
            total, allbad = 0, []
            tbad = []
            for text in disassemble(lib, tmp):
                funcs = parse(text)
                n, bad = check(funcs)
                total += n
                allbad += bad
                tbad += check_trans(funcs)
        print("%s: %d DPP instructions, %d hazards" % (lib, total, len(allbad)))
        for name, w, d in allbad[:40]:
            print("  %s\n    %x: %s %s\n    %x: %s %s" % (name, w[0], w[1], ", ".join(w[2]), d[0], d[1], ", ".join(d[2])))
        print("%s: %d transcendental results read in the next issue slot" % (lib, len(tbad)))
        for name, w, d in tbad[:20]:
            print("  %s\n    %x: %s %s\n    %x: %s %s" % (name, w[0], w[1], ", ".join(w[2]), d[0], d[1], ", ".join(d[2])))
        if allbad or tbad:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
