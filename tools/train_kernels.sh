#!/bin/bash
# Developer tool (GPU box): per-kernel GPU time of one training step (rocprofv3 --kernel-trace --stats on bench c4).
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TIER="${1:-bf16}"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_tk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_tk -- python $REPO/bench.py --workload c4 --tier $TIER --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob('/tmp/rp_tk/**/*kernel_stats.csv',recursive=True)[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("GPU ms per step %.3f   launches/step %.0f" % (tot/12e6, sum(int(r['Calls']) for r in rows)/12))
for r in rows[:10]:
    print("%9.1f us/step  calls/step %6.1f  %s" % (float(r['TotalDurationNs'])/12e3, int(r['Calls'])/12, r['Name'][:90]))
PY
