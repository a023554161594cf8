#!/bin/bash
# Build libdfanerf.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
#   build.sh          incremental: an object is rebuilt when the HASH of what it is made from changes
#   build.sh --clean  from scratch
# An object's stamp = sha256 of its own source, every header of csrc/ and include/, the compiler flags and hipcc's version.
# (Timestamps would reuse a stale object whenever a checkout puts an older header next to a newer .o: the objects travel
# with the tree to the GPU box, the file times do not mean anything there.)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/dfanerf/libdfanerf.so"
OBJ="$HERE/build"
if [ "$1" = "--clean" ]; then rm -rf "$OBJ" "$OUT"; fi
# DFN_EXTRA_FLAGS is a developer hook (timing / ablation switches: csrc/dfn_devguard.h): one stray -D would build a product
# library that renders garbage at full speed.  Extra flags need DFN_DEV_BUILD=1, which marks the library (dfn_version()
# ends in " DEV") so that dfanerf._lib refuses it on the in-tree path.
if [ -n "$DFN_EXTRA_FLAGS" ]; then
  if [ "$DFN_DEV_BUILD" != "1" ]; then
    echo "build.sh: DFN_EXTRA_FLAGS='$DFN_EXTRA_FLAGS' without DFN_DEV_BUILD=1: refusing to build the product library with developer switches (tools/build_variant.sh builds variant libraries next to it)" >&2
    exit 2
  fi
  DFN_EXTRA_FLAGS="$DFN_EXTRA_FLAGS -DDFN_DEV_BUILD=1"
fi
mkdir -p "$OBJ"
# -pragma-unroll-threshold: the MLP bodies MUST unroll completely (every fragment index, ring slot and recorder dword is a
# compile-time constant by construction; a loop left rolled puts the fragment ring and the operand vectors into scratch
# memory) and with the MX-fp8 recorder some bodies exceed LLVM's default budget of 16384 instructions
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -mllvm -pragma-unroll-threshold=200000 -I$SRC -I$HERE/../include $DFN_EXTRA_FLAGS"
HDR_HASH="$( (cat "$SRC"/*.h "$HERE"/../include/*.h; echo "$FLAGS" | sed "s#$HERE#.#g"; hipcc --version 2>/dev/null | head -2) | sha256sum | cut -d' ' -f1)"
UNITS="dfn_render dfn_render_f32 dfn_render_bf16 dfn_render_bf16e dfn_render_f16 dfn_misc dfn_api dfn_train dfn_bwd_bf16 dfn_wgrad_bf16 dfn_signal"
# the library's own stamp (next to the .so: it travels with it to the GPU box, the object directory does not): everything it
# is made from, hashed - an up-to-date library is not rebuilt
LIB_HASH="$( (echo "$HDR_HASH"; cat "$SRC"/*.hip "$SRC"/*.cpp) | sha256sum | cut -d' ' -f1)"
if [ -f "$OUT" ] && [ "$(cat "$OUT.stamp" 2>/dev/null)" = "$LIB_HASH" ]; then echo "up to date: $OUT"; exit 0; fi
rm -f "$OUT.stamp"
pids=()
for f in $UNITS; do
  ( want="$HDR_HASH $(sha256sum < "$SRC/$f.hip" | cut -d' ' -f1)"
    if [ ! -f "$OBJ/$f.o" ] || [ "$(cat "$OBJ/$f.stamp" 2>/dev/null)" != "$want" ]; then
      rm -f "$OBJ/$f.stamp"
      EXTRA=""; case "$f" in dfn_render_*|dfn_train|dfn_bwd_bf16) EXTRA="--save-temps=obj";; esac     # keep the ISA of the MLP kernels for the checks below
      hipcc $FLAGS $EXTRA -c "$SRC/$f.hip" -o "$OBJ/$f.o"
      echo "$want" > "$OBJ/$f.stamp"
    fi ) &
  pids+=($!)
done
g++ -O2 -std=c++17 -fPIC -I"$SRC" -I"$HERE/../include" -c "$SRC/dfn_plan.cpp" -o "$OBJ/dfn_plan.o"
for p in "${pids[@]}"; do wait $p; done
# the asm fragment fetch (DFN_ASM_FETCH) is only safe if nothing touches an in-flight destination register
for t in bf16 bf16e f16; do
  ISA="$OBJ/dfn_render_$t-hip-amdgcn-amd-amdhsa-gfx950.s"
  if [ -f "$ISA" ]; then
    python3 "$HERE/../tools/check_inflight.py" "$ISA" || { echo "build.sh: in-flight register hazard in the $t render kernels" >&2; exit 1; }
    # ... and the 16-bit inference kernels must not use scratch memory at all (stack objects, spilled VGPRs)
    python3 "$HERE/../tools/check_scratch.py" "$ISA" || { echo "build.sh: scratch memory in the $t inference kernels" >&2; exit 1; }
  fi
done
# the recorders' hand-written stores (dfn_mlp.h DFN_GSTORE): the hazards hipcc does not see inside inline asm
for ISA in "$OBJ"/*-hip-amdgcn-amd-amdhsa-gfx950.s; do
  [ -f "$ISA" ] && { python3 "$HERE/../tools/check_asm_stores.py" "$ISA" || { echo "build.sh: hazard at a hand-written store in $ISA" >&2; exit 1; }; }
done
rm -f "$OBJ"/*-hip-amdgcn-*.o "$OBJ"/*.hipi "$OBJ"/*.bc "$OBJ"/*.out "$OBJ"/*.resolution.txt "$OBJ"/*.hipfb "$OBJ"/*-host-*.s      # --save-temps leftovers (the device ISA stays)
OBJS=""; for f in $UNITS; do OBJS="$OBJS $OBJ/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS "$OBJ/dfn_plan.o"
echo "$LIB_HASH" > "$OUT.stamp"
echo "built $OUT"
