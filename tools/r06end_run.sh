#!/bin/bash
# Round 6, GPU session K / end (developer tool): the whole GPU suite + smoke + the driver's bench call at HEAD
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06end; mkdir -p $OUT
( time python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|Error" | tail -8 ) > $OUT/gpu_suite.txt 2>&1
cat $OUT/gpu_suite.txt | tail -6
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -9 > $OUT/smoke.txt; tail -2 $OUT/smoke.txt
( time python bench.py --steps 20 --warmup 3 ) > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06end/bench_driver_args.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype')}, 'frac', d['roofline']['frac'])
cb=d['cpu_baseline']; print('cpu', cb['value'], cb.get('multi_process',{}).get('repetition_rates'))
print('f16_range', d.get('f16_range'))
for k,v in d['other_workloads'].items(): print(k, v.get('ms_per_step'), v.get('roofline',{}).get('frac'), v.get('error'))
P
tail -4 $OUT/bench_driver_args.err
