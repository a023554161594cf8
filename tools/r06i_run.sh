#!/bin/bash
# Round 6, GPU session I (developer tool): new tests (listener training, N_samples 32 / 128), bench f16_range
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06i; mkdir -p $OUT
{
python -m pytest tests/test_gpu_train.py -q -x -s -k "listener or other_sample_counts" 2>&1 | grep -v amdgpu.ids | tail -40
python -m pytest tests/test_gpu_parity.py -q -x -s -k "other_sample_counts or coarse_f32_vs_reference or hierarchical_f32" 2>&1 | grep -v amdgpu.ids | tail -25
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['f16_range'])"
} 2>&1 | tee $OUT/log.txt
