#!/bin/bash
# Round 6, GPU session N (developer tool): compiler scheduling switch on the f32 kernels (-amdgpu-schedule-relaxed-occupancy), A/B
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06n; mkdir -p $OUT
B="python bench.py --workload c4 --tier f32 --steps 100 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
{
for r in 1 2; do
  echo -n "base: "; $B 2>/dev/null | ms
  echo -n "relaxed occupancy: "; DFN_LIB=exp_libs/f32_relaxocc.so $B 2>/dev/null | ms
done
python tools/time_fwd.py f32; DFN_LIB=exp_libs/f32_relaxocc.so python tools/time_fwd.py f32
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
