#!/bin/bash
# Round 6, GPU session Y (developer tool): every recorder store of the training kernels (f32 values, MX-fp8 / MX-fp4 tiles, ReLU
# bits) in the SGPR-base form, by hand (dfn_mlp.h DFN_GSTORE), against the library at HEAD.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06y; mkdir -p $OUT
{
echo "saddr2 tests: $(DFN_LIB=exp_libs/saddr2.so python -m pytest tests/test_gpu_train.py tests/test_gpu_wgrad.py -x -q 2>&1 | tail -1)"
for r in 1 2; do
for v in base saddr2; do echo "$v fwd f32: $(DFN_LIB=exp_libs/$v.so python tools/time_fwd.py f32 2>&1 | tail -1)"; done
for v in base saddr2; do echo "$v dx f32: $(DFN_LIB=exp_libs/$v.so python tools/time_dx.py f32 2>&1 | tail -1)"; done
for v in base saddr2; do echo "$v fwd bf16: $(DFN_LIB=exp_libs/$v.so python tools/time_fwd.py bf16 2>&1 | tail -1)"; done
for v in base saddr2; do echo "$v dx bf16: $(DFN_LIB=exp_libs/$v.so python tools/time_dx.py bf16 2>&1 | tail -1)"; done
done
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2 3; do
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
for v in base saddr2; do echo "step f32 $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
B="python bench.py --workload c4 --steps 400 --warmup 20 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
for v in base saddr2; do echo "step 16-bit $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
B="python bench.py --workload c4h --steps 200 --warmup 20 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
for v in base saddr2; do echo "step 16-bit hier $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
