#!/bin/bash
# Developer tool (GPU box): A/B wall-clock of kernel variants, interleaved to average out clock/thermal drift.
#   tools/ab.sh <workload> <rounds> lib1.so lib2.so ...      ("intree" = the shipped library)
WL="$1"; R="$2"; shift 2
for r in $(seq $R); do
  for lib in "$@"; do
    if [ "$lib" = intree ]; then unset DFN_LIB; else export DFN_LIB="$lib"; fi
    ms=$(python bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline --no-extra --sustain-seconds 0 ${AB_ARGS:-} | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['roofline'].get('kernel_ms', d['ms_per_step']))")
    echo "round $r  $lib  $ms ms"
  done
done
