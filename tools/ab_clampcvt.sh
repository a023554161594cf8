#!/bin/bash
# interleaved A/B on one box: shipping library vs the DFN_EXP_CLAMPCVT build, head activations scaled by 1/16 in both
mkdir -p gpurun_out/r04o
export DFN_BENCH_ACT_SCALE=0.0625
ARGS="--workload c2 --steps 60 --warmup 10 --sustain-seconds 0 --no-cpu-baseline --no-extra"
for r in 1 2 3; do
  python bench.py $ARGS 2>/dev/null | grep "^{" > gpurun_out/r04o/ship_$r.json
  DFN_LIB=$PWD/exp_libs/libdfanerf_clampcvt.so python bench.py $ARGS 2>/dev/null | grep "^{" > gpurun_out/r04o/exp_$r.json
done
DFN_LIB=$PWD/exp_libs/libdfanerf_clampcvt.so python bench.py --workload c1 --steps 100 --warmup 10 --sustain-seconds 0 --no-cpu-baseline --no-extra 2>/dev/null | grep "^{" > gpurun_out/r04o/exp_c1.json
python bench.py --workload c1 --steps 100 --warmup 10 --sustain-seconds 0 --no-cpu-baseline --no-extra 2>/dev/null | grep "^{" > gpurun_out/r04o/ship_c1.json
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04o/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); pc=d.get('parity_check',{})
        print(f.split('/')[-1], round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['clock_ghz'],3), pc.get('psnr_db'), pc.get('max_abs_rgb'), (pc.get('vs_oracle_pipeline') or {}).get('psnr_db'))
    except Exception as e: print(f, 'ERR', e)
P
