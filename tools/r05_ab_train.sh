#!/bin/bash
# Developer tool (GPU box), round 5: interleaved A/B of the training kernels' changes - the de-phased slab hand-over
# (DFN_DEPHASE_FWD / _BWD), the three-input packed maximum in the block-scale search (DFN_AMAX3), gradient vectors alternating
# by name in the dX chain (DFN_BWD_PINGPONG) and the straight-line trunk in the recording forward (DFN_TRAIN_TRUNK_UNROLL) -
# on the kernels alone (tools/time_fwd.py, tools/time_dx.py) and on the whole step (bench.py --workload c4 / c4h).
# Variant libraries (tools/build_variant.sh, VARIANT_UNITS="dfn_render_bf16 dfn_bwd_bf16"):
#   exp_libs/base.so      everything off (= round 4's kernels)      exp_libs/nodeph.so    the three trims, no de-phasing
#   exp_libs/dephonly.so  de-phasing only        (LIBS="..." overrides the list; ROUNDS_C4 / ROUNDS_C4H the rounds)
OUT="${1:-gpurun_out/r05b}"; mkdir -p "$OUT"
LIBS="${LIBS:-intree exp_libs/base.so exp_libs/nodeph.so exp_libs/dephonly.so}"
{
for r in 1 2; do
  for lib in $LIBS; do
    [ -f "$lib" ] || [ "$lib" = intree ] || continue
    if [ "$lib" = intree ]; then unset DFN_LIB; else export DFN_LIB="$lib"; fi
    echo "round $r $lib fwd: $(python tools/time_fwd.py bf16 2>/dev/null | tail -1)"
    python tools/time_dx.py bf16 2>/dev/null | grep "us per launch" | sed "s#^#round $r $lib dX: #"
  done
done
unset DFN_LIB
V=""
for lib in $LIBS; do
  if [ "$lib" = intree ]; then V="$V new"; else n=$(basename $lib .so); [ -f $lib ] && V="$V $n:DFN_LIB=$lib"; fi
done
WL=c4 ROUNDS=${ROUNDS_C4:-3} STEPS=1000 tools/ab_c4.sh $V
[ "${ROUNDS_C4H:-2}" -gt 0 ] && WL=c4h ROUNDS=${ROUNDS_C4H:-2} STEPS=300 tools/ab_c4.sh $V
} 2>&1 | tee "$OUT/ab_train.txt"
