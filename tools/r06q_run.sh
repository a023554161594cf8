#!/bin/bash
# Round 6, GPU session Q (developer tool): the convergence tests with the accuracy guard's numbers; then 100,000 steps of the 16-bit tier
# and the guard's verdict on THOSE weights ("worst-block margin on the 100k-step weights reported")
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06q; mkdir -p $OUT
python -m pytest tests/test_gpu_convergence.py -q -x -s 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|accuracy guard|f16 vs f32 on|Error" | tee $OUT/convergence_test.txt
DFN_CONV_INFERENCE_CHECK=1 python tools/convergence_long.py 100000 bf16_fp4:bf16:fp4:100 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|info = " | tee $OUT/convergence_100k_guard.txt | tail -12
