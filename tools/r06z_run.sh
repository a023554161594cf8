#!/bin/bash
# Round 6, GPU session Z (developer tool): the f32 training forward with the LDS-DMA of its weight stream as inline asm (SGPR-base
# form, counted waits) - it had asked for it all along and got the builtin.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06z; mkdir -p $OUT
{
echo "asmd32 tests: $(DFN_LIB=exp_libs/asmd32.so python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -k 'f32' 2>&1 | tail -1)"
for r in 1 2 3; do
for v in base asmd32; do echo "$v fwd f32: $(DFN_LIB=exp_libs/$v.so python tools/time_fwd.py f32 2>&1 | tail -1)"; done
done
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
for r in 1 2 3; do
for v in base asmd32; do echo "step f32 $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
