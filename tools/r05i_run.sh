#!/bin/bash
# Developer tool (GPU box), round 5, evidence session: the driver-style bench line, rocprofv3 summaries of the headline and of the
# training step (stats + PMC passes, tools/profile.sh), the step's timeline, the training CLI end to end, the whole GPU suite.
OUT=$PWD/gpurun_out/r05i; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
tools/profile.sh r05_c2_f16 --workload c2 > $OUT/profile_c2.txt 2>&1
tools/profile.sh r05_c4_bf16 --workload c4 > $OUT/profile_c4.txt 2>&1
tools/timeline.sh r05_c4 > $OUT/timeline_c4.txt 2>&1
(timeout 900 python tools/train_cli_timing.py 2000) > $OUT/train_cli_timing.json 2> $OUT/train_cli_timing.err
export OMP_NUM_THREADS=8
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $OUT/gpu_suite.txt
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05i/bench_driver_args.json") if l.startswith("{")][-1])
r = d["roofline"]
print("c2:", round(d["ms_per_step"], 3), "ms frac", round(r["frac"], 4), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k.startswith("ceiling") or k == "frac_of_mix"})
print("f16_range:", d.get("f16_range"))
c = d["cpu_baseline"]; print("cpu:", round(c["value"], 1), c["cores"], "single", round(c["single_process_value"], 1), c.get("multi_process_runs"))
for k, v in d["other_workloads"].items():
    print(k, v.get("ms_per_step"), v.get("dtype"), round(v["roofline"]["frac"], 4) if "roofline" in v else v)
PY
tail -8 $OUT/gpu_suite.txt; tail -3 $OUT/timeline_c4.txt; cat $OUT/train_cli_timing.json | tail -3
