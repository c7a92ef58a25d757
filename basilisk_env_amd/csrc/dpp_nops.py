#!/usr/bin/env python3
"""Build step: pad DPP read-after-write hazards in the device assembly of bsk_kernels.hip.

usage: dpp_nops.py IN.s OUT.s        (prints how many s_nop it inserted)

The kernels feed wave-uniform constants to their FMAs through `v_fmac_f64_dpp ... row_newbcast:N` / `v_mov_b64_dpp`
(bsk_device.hpp: fmac_k, get_k), written as inline asm: the compiler has no fp64 DPP FMA to select (its DPP combiner leaves
`llvm.amdgcn.update.dpp.f64` + fma as two instructions).  gfx950 wants TWO wait states between a VALU write of a VGPR and
a DPP read of it, FIVE after a VALU write of EXEC; the compiler's hazard recognizer inserts them for the instructions it
selected itself and does not look inside inline asm.  Wherever the register allocator reloads a table row (from an AGPR, a
copy, a spill) right in front of one of these instructions the DPP read would see the stale register.  So the Makefile
compiles the device side to assembly, this script inserts exactly the `s_nop`s the hazard recognizer would have, and the
result is assembled, linked and bundled as hipcc would have done it (hipcc_dpp.py).  tools/dpp_hazard.py makes the same analysis
on the disassembly of the built library (tests/test_dpp_hazard.py): the two must agree that nothing is left.

WHICH reads are padded.  LLVM's GCNHazardRecognizer::checkDPPHazards is more conservative than this pass: it pads
every VGPR *use* of a DPP instruction - also the accumulator and the plain second factor of `v_fmac_f64_dpp acc, tab, x` - against
any def.  This pass pads the operand that goes through the DPP crossbar (src0: `tab`, the row the broadcast reads another
lane's copy of) and nothing else, because that is the read the hardware performs early: the DPP permute of src0 is set up one
stage ahead of the ordinary operand fetch, which is why a VALU result needs two wait states before a DPP instruction may take
it as src0 while acc / src1 are fetched with every other VALU operand through the bypass network (an accumulator chain
`v_fmac_f64_dpp v[2:3], ..` -> `v_fmac_f64_dpp v[2:3], ..` back to back is what the harmonics walk issues 50 000 times per step;
tools/micro/dpp_rate.hip measures it at full rate with bit-exact sums, and the round-4 failure this pass was written for was a
stale *src0*).  Padding acc / src1 as well would put an s_nop pair on every broadcast FMA of the hot loops (measured cost of an
issue slot with one wave per SIMD: DESIGN.md section 4).  The checker shares this model on purpose - it answers "did the pass
do what it says on what ships" - and the model itself is held by the parity suite: every kernel with DPP operands is compared
with the CPU oracle at 1e-11 on the GPU (tests/test_gpu_*.py), where a stale accumulator would not survive.
Two-destination VALU operations (v_swap_b32, v_permlane16/32_swap) write BOTH their operands: counted as writers of both.
"""
import re
import sys

DPP_VGPR_WAIT, DPP_EXEC_WAIT = 2, 5


def regs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return None


def split_ops(s):
    return [t.strip() for t in re.split(r",(?![^\[]*\])", s)] if s else []


def process(lines):
    """-> (new lines, number of s_nop inserted, number of DPP instructions seen)"""
    # instruction table of the whole file: (line index, mnemonic, operands); labels -> index of the next instruction
    ins, label_at, pending = [], {}, []
    for li, raw in enumerate(lines):
        s = raw.split(";")[0].strip()
        if not s:
            continue
        m = re.fullmatch(r"([A-Za-z_.$][\w.$]*):", s)
        if m:
            pending.append(m.group(1))
            continue
        if s.startswith("."):
            continue
        if not raw.startswith("\t") and not raw.startswith(" "):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        if not re.match(r"^[a-z]+_", op):
            continue
        for lb in pending:
            label_at[lb] = len(ins)
        pending = []
        ins.append((li, op, split_ops(parts[1]) if len(parts) > 1 else []))
    preds = {}
    for i, (li, op, ops) in enumerate(ins):
        if (op.startswith("s_cbranch") or op == "s_branch") and ops and ops[-1] in label_at:
            preds.setdefault(label_at[ops[-1]], []).append(i)
    extra = {}          # instruction index -> wait states inserted in front of it

    def shortfall(i, src):
        """the largest number of missing wait states over every path into ins[i]"""
        worst, seen, stack = 0, set(), [(i, extra.get(i, 0))]
        while stack:
            j, ws = stack.pop()
            if ws >= DPP_EXEC_WAIT:
                continue
            for p in preds.get(j, []):          # arrived by a taken branch: the branch is one wait state
                if (p, ws + 1) not in seen:
                    seen.add((p, ws + 1))
                    stack.append((p, ws + 1))
            if j == 0:
                continue
            k = j - 1
            li, op, ops = ins[k]
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                continue
            if op.startswith("v_") and ops:
                w = regs(ops[0])
                if (op.startswith("v_swap_") or "permlane16_swap" in op or "permlane32_swap" in op) and len(ops) > 1:
                    w = (w or set()) | (regs(ops[1]) or set())          # both operands are destinations
                if w and (w & src) and ws < DPP_VGPR_WAIT:
                    worst = max(worst, DPP_VGPR_WAIT - ws)
                if op.startswith("v_cmpx") and ws < DPP_EXEC_WAIT:
                    worst = max(worst, DPP_EXEC_WAIT - ws)
            step = (int(ops[0], 0) + 1) if op == "s_nop" else 1
            step += extra.get(k, 0)
            if ws + step < DPP_EXEC_WAIT and (k, ws + step) not in seen:
                seen.add((k, ws + step))
                stack.append((k, ws + step))
        return worst

    n_dpp = 0
    for i, (li, op, ops) in enumerate(ins):
        if "_dpp" not in op:
            continue
        n_dpp += 1
        src = regs(ops[1].split()[0]) if len(ops) > 1 else None
        if not src:
            continue
        need = shortfall(i, src)
        if need > 0:
            extra[i] = extra.get(i, 0) + need
    out, at = [], {ins[i][0]: n for i, n in extra.items()}
    for li, raw in enumerate(lines):
        if li in at:
            out.append("\ts_nop %d\t; dpp_nops.py: DPP read of a register written %d instruction(s) before" % (at[li] - 1, DPP_VGPR_WAIT - at[li] if at[li] <= DPP_VGPR_WAIT else 0))
        out.append(raw)
    return out, len(extra), n_dpp


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    out, n, n_dpp = process(lines)
    open(dst, "w").write("\n".join(out))
    print("dpp_nops: %d DPP instructions, %d s_nop inserted" % (n_dpp, n))


if __name__ == "__main__":
    main()
