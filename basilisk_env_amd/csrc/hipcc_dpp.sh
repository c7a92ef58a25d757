#!/bin/bash
# hipcc -c for bsk_kernels.hip with the DPP hazard pass (dpp_nops.py) between the device compiler and the assembler:
#   device side -> assembly -> s_nop padding -> code object -> bundle;  host side compiled against that bundle.
# The steps are hipcc's own (hipcc -### -c): only the padding is added.
# usage: hipcc_dpp.sh OUT.o [compiler flags ...]        (run in csrc/; HIPCC, ARCH from the environment or the defaults below)
set -e
OUT=$1; shift
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${ARCH:-gfx950}
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
HERE=$(cd "$(dirname "$0")" && pwd)
T=$(mktemp -d /tmp/bsk_dpp.XXXXXX)
trap 'rm -rf "$T"' EXIT
$HIPCC "$@" --cuda-device-only -S $HERE/bsk_kernels.hip -o $T/dev.s
python3 $HERE/dpp_nops.py $T/dev.s $T/fix.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$ARCH -c $T/fix.s -o $T/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.hsaco $T/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH -input=/dev/null -input=$T/dev.hsaco -output=$T/dev.hipfb
$HIPCC "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $HERE/bsk_kernels.hip -o $OUT
