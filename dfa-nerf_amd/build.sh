#!/bin/bash
# Build libdfanerf.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/dfanerf/libdfanerf.so"
OBJ="$HERE/build"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -I$SRC -I$HERE/../include"
pids=()
for f in dfn_render dfn_render_f32 dfn_render_bf16 dfn_render_f16 dfn_misc dfn_api dfn_train dfn_bwd_bf16 dfn_wgrad_bf16 dfn_signal; do
  ( if [ ! -f "$OBJ/$f.o" ] || [ -n "$(find "$SRC" "$HERE/../include" -newer "$OBJ/$f.o" \( -name '*.h' -o -name "$f.hip" \) -print -quit)" ]; then
      EXTRA=""; case "$f" in dfn_render_*) EXTRA="--save-temps=obj";; esac     # keep the ISA of the render kernels for the checks below
      hipcc $FLAGS $EXTRA -c "$SRC/$f.hip" -o "$OBJ/$f.o"
    fi ) &
  pids+=($!)
done
g++ -O2 -std=c++17 -fPIC -I"$SRC" -I"$HERE/../include" -c "$SRC/dfn_plan.cpp" -o "$OBJ/dfn_plan.o"
for p in "${pids[@]}"; do wait $p; done
# the asm fragment fetch (DFN_ASM_FETCH) is only safe if nothing touches an in-flight destination register
for t in bf16 f16; do
  ISA="$OBJ/dfn_render_$t-hip-amdgcn-amd-amdhsa-gfx950.s"
  if [ -f "$ISA" ]; then
    python3 "$HERE/../tools/check_inflight.py" "$ISA" || { echo "build.sh: in-flight register hazard in the $t render kernels" >&2; exit 1; }
    # ... and the 16-bit inference kernels must not use scratch memory at all (stack objects, spilled VGPRs)
    python3 "$HERE/../tools/check_scratch.py" "$ISA" || { echo "build.sh: scratch memory in the $t inference kernels" >&2; exit 1; }
  fi
done
rm -f "$OBJ"/*-hip-amdgcn-*.o "$OBJ"/*.hipi "$OBJ"/*.bc "$OBJ"/*.out "$OBJ"/*.resolution.txt "$OBJ"/*.hipfb "$OBJ"/*-host-*.s      # --save-temps leftovers (the device ISA stays)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/dfn_render.o "$OBJ"/dfn_render_f32.o "$OBJ"/dfn_render_bf16.o "$OBJ"/dfn_render_f16.o "$OBJ"/dfn_misc.o "$OBJ"/dfn_api.o "$OBJ"/dfn_train.o "$OBJ"/dfn_bwd_bf16.o "$OBJ"/dfn_wgrad_bf16.o "$OBJ"/dfn_signal.o "$OBJ"/dfn_plan.o
echo "built $OUT"
