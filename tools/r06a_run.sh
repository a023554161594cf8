#!/bin/bash
# Round 6, GPU session A (developer tool): price the f32 training forward's recorder (stores / ReLU bits), the f32 step at HEAD,
# its weight gradients alone, then the PMC passes of the f32 step.   gpurun -- bash tools/r06a_run.sh
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06a; mkdir -p $OUT
{
echo "== training forward, f32 (tools/time_fwd.py) =="
python tools/time_fwd.py f32
for v in f32_nostore f32_nomask f32_norec; do
  [ -f exp_libs/$v.so ] && { echo "-- $v"; DFN_LIB=exp_libs/$v.so python tools/time_fwd.py f32; }
done
python tools/time_fwd.py f32
echo "== weight gradients alone, f32 =="
python tools/time_wgrad.py f32
echo "== step =="
B="python bench.py --workload c4 --tier f32 --steps 100 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do
  echo -n "base: "; $B 2>/dev/null | ms
  echo -n "DFN_TRAIN_WGRAD_SIDE=0: "; DFN_TRAIN_WGRAD_SIDE=0 $B 2>/dev/null | ms
  echo -n "DFN_TRAIN_OVERLAP=0: "; DFN_TRAIN_OVERLAP=0 $B 2>/dev/null | ms
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
bash tools/profile.sh r06_c4_f32 --workload c4 --tier f32 > $OUT/profile.log 2>&1
tail -40 $OUT/profile.log
