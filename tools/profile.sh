#!/bin/bash
# Developer tool (run on the GPU box through gpurun): rocprofv3 evidence for bench.py.
#   tools/profile.sh <tag> [bench args...]      -> gpurun_out/profiles_<tag>/{kernel_stats.csv, pmc_*.txt}
# Kernel timing (--kernel-trace --stats) and every counter group run as SEPARATE passes (no tracing domain is
# ever combined with --pmc), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
OUT="$REPO/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --no-parity-check: the 64-ray oracle launch after the timed loops would sit in every per-kernel average
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra --no-parity-check --sustain-seconds 0 $*"
rm -rf /tmp/rp && mkdir -p /tmp/rp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp/stats -- $BENCH --steps 10 --warmup 2 > "$OUT/bench_under_stats.json" 2> /tmp/rp/stats.err
cp $(find /tmp/rp/stats -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
# per (kernel, grid size): the full-frame launches on their own row (reproduces bench.py's roofline.kernel_ms directly)
python "$REPO/tools/kernel_stats_by_grid.py" "$(find /tmp/rp/stats -name '*kernel_trace.csv' | head -1)" > "$OUT/kernel_stats_by_grid.csv" 2>/dev/null
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/rp/pmc$i -- $BENCH --steps 3 --warmup 1 > /dev/null 2> /tmp/rp/pmc$i.err
  f=$(find /tmp/rp/pmc$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python "$REPO/profiles/pmc_summary.py" "$f" > "$OUT/pmc_$i.txt"; else echo "no counter file for: $grp" > "$OUT/pmc_$i.txt"; tail -5 /tmp/rp/pmc$i.err >> "$OUT/pmc_$i.txt"; fi
done
cat "$OUT"/pmc_*.txt | grep -E "render_kernel|wgrad|mlp_bwd|no counter" | head -60
head -8 "$OUT/kernel_stats.csv"
