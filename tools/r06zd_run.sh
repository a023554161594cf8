#!/bin/bash
# Round 6, GPU session ZD (developer tool): LDS layout of the render kernels with the weight ring LAST (biases + per-wave scratch below
# 64 KiB: constant offsets fit the LDS instructions' offset field).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06zd; mkdir -p $OUT
V="${1:-ringlast}"
{
echo "$V tests: $(DFN_LIB=exp_libs/$V.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | grep -E 'passed|failed' | tail -1)"
for wl in c2 c3 c1; do
B="python bench.py --workload $wl --steps 30 --warmup 5 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
for r in 1 2 3; do for v in base $V; do echo "$wl $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))")"; done; done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log_$V.txt
