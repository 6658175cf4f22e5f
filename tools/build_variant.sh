#!/bin/bash
# Builds a tuning variant of libmpe_hip.so that differs from the in-tree build only in the f32 tile kernels:
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DMPE_ABL_NO_PAIRS]   -> tools/variants/lib_<name>.so
# (run `make -C marl_amd/csrc` first: the other translation units are taken from its build/ directory)
set -e
NAME="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/marl_amd/csrc"; V="$ROOT/tools/variants"; mkdir -p "$V/$NAME"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "$V/$NAME/mpe_tile_f32.o" "$C/mpe_tile_f32.hip"
OBJS=$(ls "$C"/build/*.o | grep -v mpe_tile_f32.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/lib_$NAME.so" $OBJS "$V/$NAME/mpe_tile_f32.o"
echo "$V/lib_$NAME.so"
