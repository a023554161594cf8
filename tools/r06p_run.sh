#!/bin/bash
# Round 6, GPU session P (developer tool): world-8 CLI soak with EIGHT hardware queues per process forced, replicas broadcast through the host (the default now)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06p; mkdir -p $OUT
{
for i in 1 2 3 4 5 6; do
  echo -n "Q=8 run $i: "; GPU_MAX_HW_QUEUES=8 timeout 600 python -m pytest tests/test_gpu_driver.py -x -q -k "several_ranks_on_one_gpu and 8" 2>&1 | grep -E "passed|failed" | tail -1
done
for i in 1 2; do
  echo -n "Q=8 bench --gpus 8 --workload all run $i: "; GPU_MAX_HW_QUEUES=8 timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -k "eight_ranks" 2>&1 | grep -E "passed|failed" | tail -1
done
} 2>&1 | tee $OUT/log.txt
