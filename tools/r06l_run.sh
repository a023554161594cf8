#!/bin/bash
# Round 6, GPU session L (developer tool): the multi-rank / CLI / bench test files (world 2 and 8 on one GPU included) three times in a row
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06l; mkdir -p $OUT
{
for i in 1 2 3; do
  python -m pytest tests/test_gpu_multirank.py tests/test_gpu_driver.py tests/test_gpu_bench.py -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED" | tail -4
done
} 2>&1 | tee $OUT/log.txt
