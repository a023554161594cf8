#!/bin/bash
# Developer tool: build a variant of libdfanerf.so with extra -D flags into exp_libs/<name>.so
# (the in-tree library is untouched); select it at run time with DFN_LIB=exp_libs/<name>.so.
#   tools/build_variant.sh timing -DDFN_TIMING
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
SRC="$ROOT/dfa-nerf_amd/csrc"
OBJ="$ROOT/exp_libs/obj_$NAME"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -I$SRC -I$ROOT/include $*"
pids=()
for f in dfn_render dfn_misc dfn_api dfn_train dfn_bwd_bf16 dfn_wgrad_bf16 dfn_signal; do
  ( hipcc $FLAGS -c "$SRC/$f.hip" -o "$OBJ/$f.o" 2>"$OBJ/$f.log" ) &
  pids+=($!)
done
g++ -O2 -std=c++17 -fPIC -I"$SRC" -I"$ROOT/include" -c "$SRC/dfn_plan.cpp" -o "$OBJ/dfn_plan.o"
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/exp_libs/$NAME.so" "$OBJ"/*.o
rm -rf "$OBJ"
echo "built exp_libs/$NAME.so"
