#!/usr/bin/env python3
"""hipcc -c for bsk_kernels.hip with the DPP hazard pass (dpp_nops.py) between the device compiler and the assembler:

    device side -> assembly -> s_nop padding -> code object -> link -> bundle;  host side compiled against that bundle.

usage: hipcc_dpp.py OUT.o [compiler flags ...]        (HIPCC, ARCH from the environment or the defaults below)

The link and bundle steps are NOT spelled out here: they are taken from what `hipcc -### -c` of the running toolchain says it
would do (its `lld` and `clang-offload-bundler` command lines, with only the input / output files replaced), so that a ROCm
release which changes a flag, a target triple or the bundle alignment changes this build with it instead of silently parting
ways with plain `hipcc -c`.  If those two commands cannot be found in the driver's output the build fails - loudly.
tests/test_dpp_build.py builds the kernels both ways and holds the two results to "same kernel descriptors, same instruction
streams but for the inserted s_nop".
"""
import os
import re
import shlex
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "bsk_kernels.hip")


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def driver_plan(hipcc, flags, src=SRC):
    """-> (lld argv, bundler argv) exactly as `hipcc -### -c` prints them for these flags."""
    res = subprocess.run([hipcc] + flags + ["-###", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    lld = bundler = None
    for line in res.stderr.splitlines():
        line = line.strip()
        if not line.startswith('"'):
            continue
        argv = shlex.split(line)
        exe = os.path.basename(argv[0])
        if exe in ("lld", "ld.lld") and "elf64_amdgpu" in argv:
            lld = argv
        elif exe == "clang-offload-bundler":
            bundler = argv
    if res.returncode != 0 or not lld or not bundler:
        sys.stderr.write(res.stderr[-2000:])
        raise SystemExit("hipcc_dpp.py: could not read the device link / bundle steps from `%s -###` (toolchain changed?): lld %s, bundler %s"
                         % (hipcc, "found" if lld else "MISSING", "found" if bundler else "MISSING"))
    return lld, bundler


def relink(lld, obj, out):
    """hipcc's own lld command line with its temporary input object and output replaced by ours."""
    argv, k, seen_in = [], 0, False
    while k < len(lld):
        a = lld[k]
        if a == "-o":
            argv += ["-o", out]
            k += 2
            continue
        if a.endswith(".o") and not a.startswith("-"):
            if seen_in:
                raise SystemExit("hipcc_dpp.py: more than one input object in hipcc's device link step: %r" % lld)
            argv.append(obj)
            seen_in = True
        else:
            argv.append(a)
        k += 1
    if not seen_in:
        raise SystemExit("hipcc_dpp.py: no input object in hipcc's device link step: %r" % lld)
    return argv


def rebundle(bundler, hsaco, out):
    """hipcc's own bundler command line: host slot stays /dev/null, the device input and the output are ours."""
    argv, n_in = [], 0
    for a in bundler:
        m = re.match(r"^(-{1,2}input=)(.*)$", a)
        if m:
            n_in += 1
            argv.append(a if m.group(2) == "/dev/null" else m.group(1) + hsaco)
        elif re.match(r"^-{1,2}output=", a):
            argv.append(a.split("=", 1)[0] + "=" + out)
        else:
            argv.append(a)
    if n_in != 2:
        raise SystemExit("hipcc_dpp.py: expected a host and one device input in hipcc's bundle step: %r" % bundler)
    return argv


def build(out, flags, hipcc, arch, keep=None, pad=True):
    """``pad=False``: the same pipeline without the padding pass (tests: what the hand-made steps alone change - nothing)."""
    sys.path.insert(0, HERE)
    import dpp_nops
    llvm = os.environ.get("LLVM") or os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin")
    lld, bundler = driver_plan(hipcc, flags)
    with tempfile.TemporaryDirectory(prefix="bsk_dpp.") as t:
        dev_s, fix_s, dev_o, hsaco, fb = (os.path.join(t, n) for n in ("dev.s", "fix.s", "dev.o", "dev.hsaco", "dev.hipfb"))
        run([hipcc] + flags + ["--cuda-device-only", "-S", SRC, "-o", dev_s])
        lines = open(dev_s).read().split("\n")
        if pad:
            fixed, n, n_dpp = dpp_nops.process(lines)
        else:
            fixed, n, n_dpp = lines, 0, sum(1 for l in lines if "_dpp" in l.split(";")[0])
        open(fix_s, "w").write("\n".join(fixed))
        summary = "dpp_nops: %d DPP instructions, %d s_nop inserted" % (n_dpp, n)
        print(summary, flush=True)
        with open(out + ".dpp_nops.txt", "w") as f:       # (what __graft_entry__.build() shows when the object is up to date)
            f.write(summary + "  [flags: %s]\n" % " ".join(flags))
        run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + arch, "-c", fix_s, "-o", dev_o])
        run(relink(lld, dev_o, hsaco))
        run(rebundle(bundler, hsaco, fb))
        run([hipcc] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", SRC, "-o", out])
        if keep:                                  # (tests / inspection: the assembly before and after the pass, the code object)
            os.makedirs(keep, exist_ok=True)
            for f in (dev_s, fix_s, hsaco):
                os.replace(f, os.path.join(keep, os.path.basename(f)))
    return n_dpp, n


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    out, flags = sys.argv[1], sys.argv[2:]
    build(out, flags, os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), os.environ.get("ARCH", "gfx950"), keep=os.environ.get("BSK_DPP_KEEP"),
          pad=os.environ.get("BSK_DPP_NO_PAD") != "1")


if __name__ == "__main__":
    main()
