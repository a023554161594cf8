#!/bin/bash
# Round 6, GPU session O (developer tool): the f32 step with the torso's dX chain first (the head's GEMMs then close the step)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06o; mkdir -p $OUT
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
{
for r in 1 2 3; do
  echo -n "base: "; $B 2>/dev/null | ms
  echo -n "DFN_TRAIN_TORSO_FIRST=1: "; DFN_TRAIN_TORSO_FIRST=1 $B 2>/dev/null | ms
done
} 2>&1 | tee $OUT/log.txt
