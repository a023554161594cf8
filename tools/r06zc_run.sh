#!/bin/bash
# Round 6, GPU session ZC (developer tool): software-pipelined tile pairs in the f32 tier (layer_pipe32: a second set of accumulators,
# the previous pair's epilogue + recorder and the next pair's bias reads between the MFMAs of the pair in between).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06zc; mkdir -p $OUT
V="${1:-pipe32}"
{
echo "$V tests: $(DFN_LIB=exp_libs/$V.so python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -k 'f32' 2>&1 | grep -E 'passed|failed' | tail -1)"
for r in 1 2 3; do
for v in base $V; do echo "$v fwd f32: $(DFN_LIB=exp_libs/$v.so python tools/time_fwd.py f32 2>&1 | tail -1)"; done
done
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
for r in 1 2 3; do
for v in base $V; do echo "step f32 $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
done
B="python bench.py --workload c2 --tier f32 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --sustain-seconds 0"
for r in 1 2; do for v in base $V; do echo "c2_f32 $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('parity_check',{}).get('psnr_db'))")"; done; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log_$V.txt
