#!/bin/bash
# Round 6, GPU session R (developer tool): 256 x 256 weight-gradient kernel with the waves as a 2 x 2 grid (4 x 4 tiles per wave: 8 operand
# reads per 8-point group instead of 10) against the 4 x 1 grid (2 x 8 tiles per wave)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06r; mkdir -p $OUT
{
DFN_LIB=exp_libs/wfull_rg2.so python -m pytest tests/test_gpu_wgrad.py -x -q 2>&1 | tail -1
for r in 1 2 3; do
python tools/time_wgrad.py f32
DFN_LIB=exp_libs/wfull_rg2.so python tools/time_wgrad.py f32
done
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do echo -n "step 4x1: "; $B 2>/dev/null | ms; echo -n "step 2x2: "; DFN_LIB=exp_libs/wfull_rg2.so $B 2>/dev/null | ms; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
