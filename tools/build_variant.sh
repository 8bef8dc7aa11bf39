#!/bin/bash
# usage: tools/build_variant.sh NAME SRC.hip "-DFLAG=.. ..."  -> csrc/exp/libsavit_NAME.so (experiment builds; git-ignored)
# The ONLY build that defines SAVIT_EXPERIMENTS: ablation tiles 101-110, SAVIT_GEMM_TILE / SAVIT_PP_* / SAVIT_WGRAD_* / SAVIT_THF_DEBUG /
# SAVIT_ATTN_GENERAL environment switches exist in these libraries and not in the product libsavit.so.
set -e
cd "$(dirname "$0")/../self-attention-experiments-vision_amd/csrc"
mkdir -p exp
NAME=$1; SRC=$2; FLAGS=$3
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -ffast-math -fno-finite-math-only -DSAVIT_EXPERIMENTS $FLAGS -c $SRC -o exp/${SRC%.hip}_$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libsavit_$NAME.so exp/${SRC%.hip}_$NAME.o $OBJS
echo built exp/libsavit_$NAME.so
