#!/bin/bash
# Developer tool: build a variant of libdfanerf.so with extra -D flags into exp_libs/<name>.so
# (the in-tree library is untouched); select it at run time with DFN_LIB=exp_libs/<name>.so.
#   tools/build_variant.sh timing -DDFN_TIMING
# Only the 16-bit render translation units are recompiled with the flags (the experiments live there); everything else
# is linked from the in-tree objects (run dfa-nerf_amd/build.sh first).  VARIANT_UNITS overrides the list.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
SRC="$ROOT/dfa-nerf_amd/csrc"
BASE="$ROOT/dfa-nerf_amd/build"
OBJ="$ROOT/exp_libs/obj_$NAME"
mkdir -p "$OBJ"
UNITS="${VARIANT_UNITS:-dfn_render_f16 dfn_render_bf16}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -mllvm -pragma-unroll-threshold=200000 -I$SRC -I$ROOT/include -DDFN_DEV_BUILD=1 $*"
pids=()
for f in $UNITS; do
  ( hipcc $FLAGS --save-temps=obj -c "$SRC/$f.hip" -o "$OBJ/$f.o" 2>"$OBJ/$f.log" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { cat "$OBJ"/*.log; exit 1; }; done
for f in $UNITS; do
  ISA="$OBJ/$f-hip-amdgcn-amd-amdhsa-gfx950.s"
  python3 "$ROOT/tools/check_inflight.py" "$ISA" | grep -v " 0 hazard" || true
  python3 "$ROOT/tools/check_scratch.py" "$ISA" | grep FAIL || true
done
# dfn_api carries dfn_version(): the variant library must say " DEV" (dfanerf._lib then only loads it through DFN_LIB)
hipcc $FLAGS -c "$SRC/dfn_api.hip" -o "$OBJ/dfn_api.o"
UNITS="$UNITS dfn_api"
OTHERS=""
for o in "$BASE"/*.o; do
  b=$(basename "$o" .o); skip=0; case "$b" in *-hip-amdgcn-*) continue;; esac
  for f in $UNITS; do [ "$b" = "$f" ] && skip=1; done
  [ $skip = 0 ] && OTHERS="$OTHERS $o"
done
LINK=""; for f in $UNITS; do LINK="$LINK $OBJ/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/exp_libs/$NAME.so" $LINK $OTHERS
cp "$OBJ"/*.s "$ROOT/exp_libs/" 2>/dev/null; rm -rf "$OBJ"
echo "built exp_libs/$NAME.so"
