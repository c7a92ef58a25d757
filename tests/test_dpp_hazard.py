"""The shipped code objects carry no DPP read-after-write hazard.

The kernels' `v_fmac_f64_dpp` / `v_mov_b64_dpp` instructions are inline asm, which the compiler's hazard recognizer does not
pad; the build pads them (basilisk_env_amd/csrc/dpp_nops.py) and this test disassembles what was built and traces every DPP
read back over its predecessors (tools/dpp_hazard.py).  No GPU needed.
"""
import importlib.util
import os
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "basilisk_env_amd", "libbskgpu.so")


def _tool():
    spec = importlib.util.spec_from_file_location("dpp_hazard", os.path.join(ROOT, "tools", "dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_no_dpp_hazard_in_the_built_library():
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    d = _tool()
    total, bad, tbad = 0, [], []
    with tempfile.TemporaryDirectory() as tmp:
        for text in d.disassemble(LIB, tmp):
            funcs = d.parse(text)
            n, b = d.check(funcs)
            total += n
            bad += b
            tbad += d.check_trans(funcs)       # (the other hazard an inline-asm consumer could hide: a transcendental's result read at once)
    assert total > 1000, "the scenario kernels' DPP instructions were not found: %d" % total
    # EVERY translation unit's code object was looked at: the library holds one offload bundle per unit and clang-offload-bundler reads
    # only the first of a file - from the eight-unit build of round 5 until round 6 this check saw one eighth of the kernels (30 840 of
    # 195 296 DPP instructions).  The padding pass leaves its count per unit beside the object it built: the sums must agree.
    import glob
    import re
    notes = sorted(glob.glob(os.path.join(ROOT, "basilisk_env_amd", "csrc", "bsk_kernels_tu*.o.dpp_nops.txt")))
    if notes:
        built = sum(int(re.match(r"dpp_nops: (\d+) DPP instructions", open(f).read()).group(1)) for f in notes)
        assert len(notes) == 8 and total >= built > 100000, (total, built, len(notes))
    assert not bad, "%d DPP hazards, first: %s" % (len(bad), bad[0])
    assert not tbad, "%d transcendental forwarding hazards, first: %s" % (len(tbad), tbad[0])


def test_the_padding_pass_finds_and_fixes_a_planted_hazard():
    spec = importlib.util.spec_from_file_location("dpp_nops", os.path.join(ROOT, "basilisk_env_amd", "csrc", "dpp_nops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src = """f:
\tv_accvgpr_read_b32 v21, a47
\tv_fmac_f64_dpp v[2:3], v[20:21], v[0:1] row_newbcast:14 row_mask:0xf bank_mask:0xf
\tv_mov_b64_e32 v[30:31], v[40:41]
\tv_add_f64 v[8:9], v[8:9], v[8:9]
\tv_mov_b64_dpp v[4:5], v[30:31] row_newbcast:0 row_mask:0xf bank_mask:0xf
\ts_cbranch_scc1 .L1
\tv_mov_b64_e32 v[50:51], 0
.L1:
\tv_fmac_f64_dpp v[6:7], v[50:51], v[0:1] row_newbcast:1 row_mask:0xf bank_mask:0xf
\tv_mov_b64_e32 v[60:61], 0
\ts_nop 1
\tv_fmac_f64_dpp v[6:7], v[60:61], v[0:1] row_newbcast:1 row_mask:0xf bank_mask:0xf
\ts_endpgm
""".split("\n")
    out, n, n_dpp = mod.process(src)
    assert n_dpp == 4 and n == 3
    text = "\n".join(out)
    # zero wait states -> s_nop 1, one wait state -> s_nop 0, behind a label on the fall-through path -> s_nop 1; already padded -> nothing
    assert text.count("s_nop 1\t; dpp_nops") == 2 and text.count("s_nop 0\t; dpp_nops") == 1
    out2, n2, _ = mod.process(out)
    assert n2 == 0
