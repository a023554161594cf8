#!/bin/bash
# Round 6, GPU session T (developer tool): do a narrow workgroup and a 256 x 256 workgroup really share a compute unit?  Kernel
# timeline of the weight-gradient call alone with the forked narrow launch, and the step with eight hardware queues.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out/r06t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
{
for v in half2 half2fork; do
rm -rf /tmp/tl_$v
DFN_LIB=$REPO/exp_libs/$v.so rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -- python $REPO/tools/time_wgrad.py f32 > /tmp/tl_$v.log 2>&1
f=$(find /tmp/tl_$v -name "*kernel_trace.csv" | head -1)
echo "== $v"
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("dfn::", ""), r.get("Queue_Id", "")) for r in rows)
ev = [e for e in ev if "wgrad" in e[2] or "reduce" in e[2]]
tail = ev[-24:-12]
t0 = tail[0][0]
for s, e, n, q in tail: print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n}")
P
done
cd $REPO
B="python bench.py --workload c4 --tier f32 --steps 150 --warmup 10 --no-extra --no-cpu-baseline --sustain-seconds 0 --no-parity-check"
ms() { python -c "import sys,json; print('%.4f ms' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do
echo "step half2 (lib) q4: $(DFN_LIB=exp_libs/half2.so $B 2>/dev/null | ms)"
echo "step half2 q8: $(GPU_MAX_HW_QUEUES=8 DFN_LIB=exp_libs/half2.so $B 2>/dev/null | ms)"
echo "step half2fork q8: $(GPU_MAX_HW_QUEUES=8 DFN_LIB=exp_libs/half2fork.so $B 2>/dev/null | ms)"
echo "step half2fork head-only-off q4: $(DFN_WGRAD_FORK=0 DFN_LIB=exp_libs/half2fork.so $B 2>/dev/null | ms)"
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
