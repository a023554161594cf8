#!/bin/bash
# Round 6, GPU session J (developer tool): schedule switches of the f32 step, interleaved
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06j; mkdir -p $OUT
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
{
for r in 1 2 3; do
  echo -n "base: "; $B 2>/dev/null | ms
  echo -n "DFN_TRAIN_SIG_FIRST=0: "; DFN_TRAIN_SIG_FIRST=0 $B 2>/dev/null | ms
  echo -n "DFN_TRAIN_SPLIT_JOIN=0: "; DFN_TRAIN_SPLIT_JOIN=0 $B 2>/dev/null | ms
  echo -n "DFN_TRAIN_SIG_PRIO=0: "; DFN_TRAIN_SIG_PRIO=0 $B 2>/dev/null | ms
done
} 2>&1 | tee $OUT/log.txt
