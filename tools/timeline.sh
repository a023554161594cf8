#!/bin/bash
# Developer tool (GPU box): kernel timeline of one steady-state training step of `bench.py --workload c4`.
#   tools/timeline.sh <tag> [ENV=VAL ...]   -> gpurun_out/timeline_<tag>.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
for kv in "$@"; do export "$kv"; done
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -- python "$REPO/bench.py" --workload ${WL:-c4} ${TIER:+--tier $TIER} --steps ${STEPS:-30} --warmup 10 \
    --no-cpu-baseline --no-extra --sustain-seconds 0 > /tmp/tl_$TAG.log 2>&1
f=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python "$REPO/tools/timeline.py" "$f" -5 > "$REPO/gpurun_out/timeline_$TAG.txt"
tail -1 "$REPO/gpurun_out/timeline_$TAG.txt"
