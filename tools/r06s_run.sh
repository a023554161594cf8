#!/bin/bash
# Round 6, GPU session S (developer tool): the narrow weight-gradient GEMMs BESIDE the 256 x 256 ones on the same compute units.
# The 256 x 256 kernel takes 16-point steps (two 32-KiB stages = 64 KiB instead of 128) so that a narrow workgroup (72 KiB,
# 124 registers) fits the same compute unit (356 + 128 registers per SIMD), the narrow launch forks onto a stream of its own.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06s; mkdir -p $OUT
{
for v in half2 half2fork fork; do echo -n "$v tests: "; DFN_LIB=exp_libs/$v.so python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_train.py -x -q -k "wgrad or f32" 2>&1 | tail -1; done
for r in 1 2; do
echo -n "base  "; python tools/time_wgrad.py f32
for v in half2 half2fork fork; do echo -n "$v  "; DFN_LIB=exp_libs/$v.so python tools/time_wgrad.py f32; done
done
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2 3; do echo -n "step base: "; $B 2>/dev/null | ms; for v in half2 half2fork fork; do echo -n "step $v: "; DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms; done; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
