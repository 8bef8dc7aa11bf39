#!/bin/bash
# usage: tools/disasm.sh <object.o> [out.s]  - disassemble the gfx950 code object embedded in a HIP object file
set -e
OBJ=$(realpath "$1"); OUT=${2:-/dev/stdout}
T=$(mktemp -d); cp "$OBJ" $T/in.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $T/in.o > /dev/null
CO=$(ls $T | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/$CO > "$OUT"
rm -rf $T
