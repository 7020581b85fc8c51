#!/bin/bash
# Development tool: compile one .hip file with the VOP3 re-encoding post-pass (tools/e64.py) into an object.
# usage: tools/build_e64.sh <file.hip> <out.o> [extra hipcc flags]
set -e
SRC=$1; OUT=$2; shift 2
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -ffp-contract=off $@"
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
/opt/rocm/bin/hipcc $F --cuda-device-only -S $SRC -o $T/dev.s 2>/dev/null
python3 $(dirname $0)/e64.py $T/dev.s $T/e64.s
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/e64.s -o $T/dev.o
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.co $T/dev.o
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.co -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $F --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $SRC -o $OUT
rm -rf $T
