"""The assembly post-pass of the kernel build is held to "plain hipcc + s_nop, nothing else" (CPU; no GPU needed).

csrc/hipcc_dpp.py compiles the device side of bsk_kernels.hip to assembly, pads DPP read-after-write hazards the compiler does
not see inside inline asm (csrc/dpp_nops.py), and then assembles, links and bundles the result the way hipcc does - with the
link / bundle command lines read from `hipcc -###` of the running toolchain.  A ROCm point release could break either half
silently, so this builds the smallest kernel set (-DBSK_FAST_BUILD=3: the drop-in env's kernel in its three forms) BOTH ways
and checks, per kernel symbol:
  * the padded assembly is the compiler's assembly plus `s_nop` lines carrying the pass's tag - not one other line differs;
  * the hand-made pipeline WITHOUT the padding yields the code object plain `hipcc -c` embeds: same kernel metadata (VGPR / AGPR /
    SGPR counts, LDS, scratch, kernarg size), same kernel descriptors, same instruction stream;
  * the padded code object has the same metadata and descriptors, and its instruction stream is the plain one with exactly the
    reported number of `s_nop` more;
  * a hand edit of one line of the padded assembly makes the comparison fail.
"""
import importlib.util
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "basilisk_env_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-Wno-unused-function", "-DBSK_FAST_BUILD=3"]
TAG = "; dpp_nops.py"

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(LLVM + "/llvm-objdump")), reason="needs the ROCm toolchain")


def _mod(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def embedded_code_object(obj, out):
    """the gfx950 code object inside a host object `hipcc -c` wrote (.hip_fatbin section -> offload bundle -> ELF)"""
    fat = out + ".fat"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    tgt = [t for t in _run([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + fat]).split() if "gfx950" in t]
    assert len(tgt) == 1, tgt
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + tgt[0], "--input=" + fat, "--output=" + out], check=True)
    return out


def kernel_metadata(hsaco):
    """{kernel symbol: {field: value}} from the AMDGPU metadata note (what the runtime reads to launch the kernel)."""
    import yaml
    text = _run([LLVM + "/llvm-readelf", "--notes", hsaco])
    doc = text[text.index("---"):]
    doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
    meta = yaml.safe_load(doc)
    keep = ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "kernarg_segment_size",
            "kernarg_segment_align", "max_flat_workgroup_size", "wavefront_size", "vgpr_spill_count", "sgpr_spill_count", "uses_dynamic_stack")
    return {k[".symbol"]: {f: k.get("." + f) for f in keep} for k in meta["amdhsa.kernels"]}


def kernel_descriptors(hsaco):
    """{kd symbol: 64 descriptor bytes as hex, the code entry offset (bytes 16-23: it moves with every inserted s_nop) masked}"""
    syms = _run([LLVM + "/llvm-readelf", "--symbols", "--wide", hsaco])
    secs = _run([LLVM + "/llvm-readelf", "--sections", "--wide", hsaco])
    ro = re.search(r"\]\s+\.rodata\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
    addr, off = int(ro.group(1), 16), int(ro.group(2), 16)
    blob = open(hsaco, "rb").read()
    out = {}
    for line in syms.splitlines():
        m = re.match(r"^\s*\d+:\s+([0-9a-f]+)\s+64\s+OBJECT\s+\S+\s+\S+\s+\S+\s+(\S+\.kd)$", line)
        if m:
            a = int(m.group(1), 16) - addr + off
            kd = bytearray(blob[a:a + 64])
            kd[16:24] = b"\0" * 8
            out[m.group(2)] = kd.hex()
    return out


def instruction_streams(hsaco):
    """{function: [instruction text]} with pc-relative immediates (branch offsets, the literal behind s_getpc_b64) normalised."""
    text = _run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", hsaco])
    funcs, cur, since_getpc = {}, None, 99
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.strip().split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_call_b64":
            ins = op + " <pc-relative>"
        since_getpc = 0 if op == "s_getpc_b64" else since_getpc + 1
        if since_getpc in (1, 2, 3) and op in ("s_add_u32", "s_addc_u32", "s_sub_u32", "s_subb_u32"):
            ins = re.sub(r",\s*(0x[0-9a-f]+|-?\d+)\s*$", ", <pc-relative>", ins)
        cur.append(ins)
    return funcs


@pytest.fixture(scope="module")
def builds(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("dppbuild"))
    hd = _mod("hipcc_dpp", os.path.join(CSRC, "hipcc_dpp.py"))
    src = os.path.join(CSRC, "bsk_kernels.hip")
    plain_o = os.path.join(d, "plain.o")
    subprocess.run([HIPCC] + FLAGS + ["-c", src, "-o", plain_o], check=True, capture_output=True)
    n_dpp, n_nop = hd.build(os.path.join(d, "padded.o"), FLAGS, HIPCC, "gfx950", keep=os.path.join(d, "keep"))
    keep = os.path.join(d, "keep")
    # the hand-made pipeline on the UNPADDED assembly (no second compilation: assemble + hipcc's own link step)
    lld, _ = hd.driver_plan(HIPCC, FLAGS)
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", os.path.join(keep, "dev.s"), "-o", os.path.join(d, "nopad.o")], check=True)
    subprocess.run(hd.relink(lld, os.path.join(d, "nopad.o"), os.path.join(d, "nopad.hsaco")), check=True)
    return {"dir": d, "hd": hd, "plain": embedded_code_object(plain_o, os.path.join(d, "plain.hsaco")),
            "padded": embedded_code_object(os.path.join(d, "padded.o"), os.path.join(d, "padded.hsaco")),
            "padded_kept": os.path.join(keep, "dev.hsaco"), "nopad": os.path.join(d, "nopad.hsaco"),
            "dev_s": os.path.join(keep, "dev.s"), "fix_s": os.path.join(keep, "fix.s"), "n_dpp": n_dpp, "n_nop": n_nop}


def only_nops_added(dev_lines, fix_lines):
    """-> number of inserted lines when `fix` is `dev` plus tagged s_nop lines and nothing else; raises otherwise"""
    k, added = 0, 0
    for line in fix_lines:
        if k < len(dev_lines) and line == dev_lines[k]:
            k += 1
        elif re.fullmatch(r"\ts_nop \d+\t" + re.escape(TAG) + r".*", line):
            added += 1
        else:
            raise AssertionError("padded assembly differs from the compiler's beyond inserted s_nop: %r (compiler line %d: %r)" % (line, k, dev_lines[k] if k < len(dev_lines) else None))
    assert k == len(dev_lines), "padded assembly is missing compiler lines from %d on" % k
    return added


def test_padded_assembly_is_the_compilers_plus_tagged_nops(builds):
    dev, fix = open(builds["dev_s"]).read().split("\n"), open(builds["fix_s"]).read().split("\n")
    assert builds["n_dpp"] > 500 and builds["n_nop"] > 0          # the scenario kernels' broadcast FMAs, and real hazards among them
    assert only_nops_added(dev, fix) == builds["n_nop"]
    # one line edited by hand - an operand, a dropped instruction - and the comparison fails
    k = next(i for i, l in enumerate(fix) if l.startswith("\tv_fmac_f64_dpp"))
    for tampered in (fix[:k] + [fix[k].replace("row_newbcast", "row_share")] + fix[k + 1:], fix[:k] + fix[k + 1:], fix[:k] + ["\ts_nop 0"] + fix[k:]):
        with pytest.raises(AssertionError):
            only_nops_added(dev, tampered)


def test_hand_made_pipeline_without_padding_equals_plain_hipcc(builds):
    """assemble + hipcc's own link step on the compiler's unpadded assembly = the code object `hipcc -c` embeds"""
    mp, mn = kernel_metadata(builds["plain"]), kernel_metadata(builds["nopad"])
    assert len(mp) >= 3 and mp == mn
    assert kernel_descriptors(builds["plain"]) == kernel_descriptors(builds["nopad"])
    sp, sn = instruction_streams(builds["plain"]), instruction_streams(builds["nopad"])
    assert sp.keys() == sn.keys()
    for f in sp:
        assert sp[f] == sn[f], f


def test_padded_code_object_is_plain_plus_nops(builds):
    assert open(builds["padded"], "rb").read() == open(builds["padded_kept"], "rb").read()      # what the host object embeds is what was linked
    mp, md = kernel_metadata(builds["plain"]), kernel_metadata(builds["padded"])
    assert mp == md and all(int(v["vgpr_count"]) > 0 for v in mp.values())
    assert kernel_descriptors(builds["plain"]) == kernel_descriptors(builds["padded"])
    sp, sd = instruction_streams(builds["plain"]), instruction_streams(builds["padded"])
    assert sp.keys() == sd.keys()
    extra = 0
    for f in sp:
        a = [i for i in sp[f] if not i.startswith("s_nop")]
        b = [i for i in sd[f] if not i.startswith("s_nop")]
        assert a == b, f
        extra += sum(i.startswith("s_nop") for i in sd[f]) - sum(i.startswith("s_nop") for i in sp[f])
    assert extra == builds["n_nop"]


def test_link_and_bundle_steps_come_from_the_toolchain(builds):
    """the lld / clang-offload-bundler command lines are hipcc's own (`hipcc -###`), inputs and outputs replaced; a driver output
    without them is an error, not a silent fallback"""
    hd = builds["hd"]
    lld, bundler = hd.driver_plan(HIPCC, FLAGS)
    assert "elf64_amdgpu" in lld and any("gfx950" in a for a in bundler)
    rl = hd.relink(lld, "/x/in.o", "/x/out.hsaco")
    assert rl.count("/x/in.o") == 1 and rl[rl.index("-o") + 1] == "/x/out.hsaco" and [a for a in rl if a not in ("/x/in.o", "/x/out.hsaco")] == [a for k, a in enumerate(lld) if not (a.endswith(".o") and not a.startswith("-")) and not (k > 0 and lld[k - 1] == "-o")]
    rb = hd.rebundle(bundler, "/x/dev.hsaco", "/x/out.hipfb")
    assert sum(a.endswith("=/dev/null") for a in rb) == 1 and sum(a.endswith("=/x/dev.hsaco") for a in rb) == 1 and sum(a.endswith("=/x/out.hipfb") for a in rb) == 1
    with pytest.raises(SystemExit):
        hd.driver_plan("/bin/true", FLAGS)
