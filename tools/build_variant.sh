#!/bin/bash
# build a variant of the library from ONE alternative csrc source (timing-only ablations, same-box A/B of two builds):
#   tools/build_variant.sh <variant .hip path> <name of the csrc file it replaces> <output .so under eraxvif5tts_amd/lib/>
set -e
cd "$(dirname "$0")/.."
SRC=$1; REPL=$2; OUT=$3
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=off -fvisibility=hidden"
TMP=$(mktemp -d)
cp "$SRC" eraxvif5tts_amd/csrc/_variant_tmp.hip
hipcc $FLAGS -c eraxvif5tts_amd/csrc/_variant_tmp.hip -o $TMP/variant.o
rm -f eraxvif5tts_amd/csrc/_variant_tmp.hip
OBJS=$(ls eraxvif5tts_amd/build/*.o | grep -v "/${REPL%.hip}.o")
hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $TMP/variant.o -o eraxvif5tts_amd/lib/$OUT
rm -rf $TMP
echo "built eraxvif5tts_amd/lib/$OUT"
