#!/bin/bash
# Round 6, GPU session B (developer tool): the LDS kernel for the narrow weight-gradient GEMMs of the f32 tier - parity, timing, step.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06b; mkdir -p $OUT
{
python -m pytest tests/test_gpu_wgrad.py -x -q 2>&1 | tail -5
python -m pytest tests/test_gpu_train.py -x -q -k "f32 or golden or reproducible or schedules" 2>&1 | tail -5
echo "== weight gradients alone, f32 =="
python tools/time_wgrad.py f32
DFN_LIB=exp_libs/wfull1.so python tools/time_wgrad.py f32
python tools/time_wgrad.py f32
DFN_LIB=exp_libs/wfull1.so python tools/time_wgrad.py f32
echo "== step =="
B="python bench.py --workload c4 --tier f32 --steps 100 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do
  echo -n "base: "; $B 2>/dev/null | ms
  echo -n "wfull1: "; DFN_LIB=exp_libs/wfull1.so $B 2>/dev/null | ms
  echo -n "DFN_TRAIN_WGRAD_SIDE=0: "; DFN_TRAIN_WGRAD_SIDE=0 $B 2>/dev/null | ms
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --tier f32 --steps 10 --warmup 2 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check > /dev/null 2>&1
cp $(find /tmp/rp -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv
head -8 $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv | cut -c1-100,200-330
