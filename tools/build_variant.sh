#!/bin/bash
# usage: tools/build_variant.sh NAME SRC.hip "-DFLAG=.. ..."  -> csrc/exp/libsavit_NAME.so (experiment builds; git-ignored)
set -e
cd "$(dirname "$0")/../self-attention-experiments-vision_amd/csrc"
mkdir -p exp
NAME=$1; SRC=$2; FLAGS=$3
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -ffast-math -fno-finite-math-only $FLAGS -c $SRC -o exp/${SRC%.hip}_$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libsavit_$NAME.so exp/${SRC%.hip}_$NAME.o $OBJS
echo built exp/libsavit_$NAME.so
