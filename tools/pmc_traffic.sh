#!/bin/bash
# Developer tool (GPU box): HBM traffic of render_kernel for one workload: separate FETCH_SIZE / WRITE_SIZE passes.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
WL="$1"; TIER="${2:-bf16}"; SIZE="${3:-450}"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rp_$c -- python $REPO/bench.py --workload $WL --tier $TIER --size $SIZE --no-cpu-baseline --no-extra --no-parity-check --sustain-seconds 0 --steps 3 --warmup 1 > /dev/null 2>&1
  python $REPO/profiles/pmc_summary.py $(find /tmp/rp_$c -name '*counter_collection.csv' | head -1) render_kernel
done
