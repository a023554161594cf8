#!/bin/bash
# Developer tool (GPU box): PMC counters of the kernels of an arbitrary command, one --pmc pass per counter group (never
# with a tracing domain other than --kernel-trace).   tools/pmc_cmd.sh <tag> <kernel regex> -- <command...>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; RE="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"; : > "$REPO/gpurun_out/pmc_$TAG.txt"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES SQ_INST_LEVEL_LDS"; do
  i=$((i+1)); rm -rf /tmp/rp_pc$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/rp_pc$i -- "$@" > /dev/null 2> /tmp/rp_pc$i.err
  f=$(find /tmp/rp_pc$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python "$REPO/profiles/pmc_summary.py" "$f" | grep -E "$RE" >> "$REPO/gpurun_out/pmc_$TAG.txt"; else echo "no counters for: $grp" >> "$REPO/gpurun_out/pmc_$TAG.txt"; tail -3 /tmp/rp_pc$i.err >> "$REPO/gpurun_out/pmc_$TAG.txt"; fi
done
cat "$REPO/gpurun_out/pmc_$TAG.txt"
