#!/bin/bash
# Developer tool (GPU box): interleaved A/B of kernel variants with tools/time_render.py (no result checks: usable with
# deliberately wrong timing-experiment builds).   tools/ab_raw.sh <c2|c3> <tier> <rounds> lib1.so lib2.so ...
WL="$1"; TIER="$2"; R="$3"; shift 3
for r in $(seq $R); do
  for lib in "$@"; do
    if [ "$lib" = intree ]; then unset DFN_LIB; else export DFN_LIB="$lib"; fi
    echo "round $r  $(python tools/time_render.py $WL $TIER 12 2>/dev/null | tail -1)"
  done
done
