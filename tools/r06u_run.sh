#!/bin/bash
# Round 6, GPU session U (developer tool): 16-point steps of the weight-gradient kernels - stage counts, the narrow shapes too,
# and the forked narrow launch for the torso only (DFN_WGRAD_FORK = bit mask over the fields).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06u; mkdir -p $OUT
{
for v in half3 half4 half2n half2nfork; do echo "$v tests: $(DFN_LIB=exp_libs/$v.so python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_train.py -x -q -k 'wgrad or f32' 2>&1 | tail -1)"; done
for r in 1 2; do
echo "base   $(python tools/time_wgrad.py f32)"
for v in half2 half3 half4 half2n; do echo "$v  $(DFN_LIB=exp_libs/$v.so python tools/time_wgrad.py f32)"; done
echo "half2nfork  $(DFN_LIB=exp_libs/half2nfork.so python tools/time_wgrad.py f32)"
done
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2 3; do
echo "step base: $($B 2>/dev/null | ms)"
for v in half2 half3 half4 half2n; do echo "step $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
echo "step half2fork torso only: $(DFN_WGRAD_FORK=2 DFN_LIB=exp_libs/half2fork.so $B 2>/dev/null | ms)"
echo "step half2nfork torso only: $(DFN_WGRAD_FORK=2 DFN_LIB=exp_libs/half2nfork.so $B 2>/dev/null | ms)"
echo "step half2nfork head only: $(DFN_WGRAD_FORK=1 DFN_LIB=exp_libs/half2nfork.so $B 2>/dev/null | ms)"
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
