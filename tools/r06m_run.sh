#!/bin/bash
# Round 6, GPU session M (developer tool): C3 with EVERY pass of both fields software-pipelined (DFN_PIPE_TWO=1: the configuration a
# three-launch split of the two-field frame could at best reach, measured inside one kernel) against the shipping kernel, and C2, interleaved
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06m; mkdir -p $OUT
B="python bench.py --steps 60 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.3f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
{
for r in 1 2 3; do
  echo -n "c3 shipping: "; $B --workload c3 2>/dev/null | ms
  echo -n "c3 DFN_PIPE_TWO=1 (all passes pipelined, 15 spilled VGPRs): "; DFN_LIB=exp_libs/f16_pipe_two.so $B --workload c3 2>/dev/null | ms
  echo -n "c2 shipping: "; $B --workload c2 2>/dev/null | ms
done
} 2>&1 | tee $OUT/log.txt
