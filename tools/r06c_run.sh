#!/bin/bash
# Round 6, GPU session C (developer tool): f32 weight gradients with prefetched LDS operands - parity, timing, step
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06c; mkdir -p $OUT
{
python -m pytest tests/test_gpu_wgrad.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_train.py -x -q -k "f32 or golden or reproducible or schedules" 2>&1 | tail -3
python tools/time_wgrad.py f32
python tools/time_wgrad.py f32
B="python bench.py --workload c4 --tier f32 --steps 100 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2 3; do echo -n "step: "; $B 2>/dev/null | ms; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
