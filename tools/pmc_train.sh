#!/bin/bash
# Developer tool (GPU box): vector-memory / wait counters of the training kernels (bench c4), one --pmc pass per group,
# never together with a tracing domain other than --kernel-trace.     tools/pmc_train.sh [tier]
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TIER="${1:-bf16}"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VALU" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf /tmp/rp_pt$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/rp_pt$i -- python $REPO/bench.py --workload c4 --tier $TIER --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/rp_pt$i.err
  f=$(find /tmp/rp_pt$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python "$REPO/profiles/pmc_summary.py" "$f" | grep -E "render_kernel|mlp_bwd|wgrad"; else echo "no counters for: $grp"; tail -3 /tmp/rp_pt$i.err; fi
done
