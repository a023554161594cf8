#!/bin/bash
# Round 6, GPU session H (developer tool): the whole GPU suite + smoke at HEAD, counters / timeline of the f32 step, the driver's bench call
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06h; mkdir -p $OUT
( time python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15 ) > $OUT/gpu_suite.txt 2>&1
tail -5 $OUT/gpu_suite.txt
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -9 > $OUT/smoke.txt; tail -3 $OUT/smoke.txt
bash tools/profile.sh r06_c4_f32 --workload c4 --tier f32 > $OUT/profile.log 2>&1
TIER=f32 bash tools/timeline.sh r06_c4_f32 > /dev/null 2>&1; tail -1 gpurun_out/timeline_r06_c4_f32.txt
( time python bench.py --steps 20 --warmup 3 ) > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
tail -c 1500 $OUT/bench_driver_args.json; tail -4 $OUT/bench_driver_args.err
