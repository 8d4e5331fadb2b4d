#!/bin/bash
# register / scratch / LDS metadata of the kernels in a built librex .so:  profiles/somta.sh <lib.so> [grep pattern]
# (dumps the fat binary section, unbundles the gfx950 code object, prints profiles/kmeta.py's table)
set -e
L=$1; PAT=${2:-.}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$L" /dev/null 2>/dev/null || /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$L" $T/copy.so
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co
python3 "$(dirname "$0")/kmeta.py" $T/k.co | grep -E "$PAT" || true
rm -rf $T
