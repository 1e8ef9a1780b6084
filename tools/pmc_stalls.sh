#!/bin/bash
# Where do the waves wait?  Four PMC passes (each its own run, --pmc only) of one bench.py command line.
#   usage (inside gpurun): bash tools/pmc_stalls.sh TAG <bench.py arguments...>
# Afterwards, here: python tools/pmc_stalls_report.py TAG KERNEL_SUBSTRING
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline $*"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE TCC_BUSY_sum" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/st${TAG}_$i -- $B > $O/st${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/st${TAG}_$i.log; }
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st${TAG}_kt -- $B > $O/st${TAG}_kt.log 2>&1
echo done
