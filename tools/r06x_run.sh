#!/bin/bash
# Round 6, GPU session X (developer tool): the form of the f32 recorder's store instruction - SGPR base + 32-bit lane offset
# written by hand (hipcc picks a 64-bit vector address) and without the non-temporal hint.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06x; mkdir -p $OUT
{
for v in saddr plainst; do echo "$v tests: $(DFN_LIB=exp_libs/$v.so python -m pytest tests/test_gpu_train.py tests/test_gpu_wgrad.py -x -q -k 'f32 or wgrad' 2>&1 | tail -1)"; done
for r in 1 2; do
for v in base saddr plainst; do echo "$v fwd: $(DFN_LIB=exp_libs/$v.so python tools/time_fwd.py f32 2>&1 | tail -1)"; done
for v in base saddr plainst; do echo "$v dx: $(DFN_LIB=exp_libs/$v.so python tools/time_dx.py f32 2>&1 | tail -1)"; done
done
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2 3; do
for v in base saddr plainst; do echo "step $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
