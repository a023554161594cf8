#!/bin/bash
# Build libdfanerf.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/dfanerf/libdfanerf.so"
OBJ="$HERE/build"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$SRC -I$HERE/../include"
pids=()
for f in dfn_render dfn_misc dfn_api dfn_train dfn_signal; do
  ( if [ ! -f "$OBJ/$f.o" ] || [ -n "$(find "$SRC" "$HERE/../include" -newer "$OBJ/$f.o" \( -name '*.h' -o -name "$f.hip" \) -print -quit)" ]; then
      hipcc $FLAGS -c "$SRC/$f.hip" -o "$OBJ/$f.o"
    fi ) &
  pids+=($!)
done
g++ -O2 -std=c++17 -fPIC -I"$SRC" -I"$HERE/../include" -c "$SRC/dfn_plan.cpp" -o "$OBJ/dfn_plan.o"
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/dfn_render.o "$OBJ"/dfn_misc.o "$OBJ"/dfn_api.o "$OBJ"/dfn_train.o "$OBJ"/dfn_signal.o "$OBJ"/dfn_plan.o
echo "built $OUT"
