#!/bin/bash
# Round 6, GPU session W (developer tool, timing only - results may be wrong): what do the recorder's stores in the in-order
# vector-memory queue cost at the weight slab hand-over?  s_waitcnt vmcnt(8) there also waits for every store older than the
# eight youngest operations; DFN_EXP_VMCNT=24 / 40 lets the stores of the last slab period stay in flight.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06w; mkdir -p $OUT
{
for r in 1 2; do
for v in base vm24 vm40; do echo "$v fwd: $(DFN_LIB=exp_libs/$v.so python tools/time_fwd.py f32 2>&1 | tail -1)"; done
for v in base vm24 vm40; do echo "$v dx: $(DFN_LIB=exp_libs/$v.so python tools/time_dx.py f32 2>&1 | tail -1)"; done
done
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do
for v in base vm24 vm40; do echo "step $v: $(DFN_LIB=exp_libs/$v.so $B 2>/dev/null | ms)"; done
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
