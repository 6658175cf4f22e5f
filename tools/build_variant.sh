#!/bin/bash
# Builds a tuning variant of libmpe_hip.so that differs from the in-tree build in ONE translation unit:
#   tools/build_variant.sh <name> <unit, e.g. mpe_tile_f32> [extra hipcc flags, e.g. -DMPE_ABL_NO_PAIRS]
#   -> tools/variants/lib_<name>.so      (use with MARL_AMD_LIB=... or tools/run_variants.sh)
# (run `make -C marl_amd/csrc` first: the other translation units are taken from its build/ directory)
set -e
NAME="$1"; UNIT="$2"; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/marl_amd/csrc"; V="$ROOT/tools/variants"; mkdir -p "$V/$NAME"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o "$V/$NAME/$UNIT.o" "$C/$UNIT.hip"
OBJS=$(ls "$C"/build/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/lib_$NAME.so" $OBJS "$V/$NAME/$UNIT.o"
echo "$V/lib_$NAME.so"
