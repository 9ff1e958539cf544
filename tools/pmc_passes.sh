#!/bin/bash
# Separate rocprofv3 --pmc passes (one small counter set each, every pass under its own timeout:
# a counter set the hardware cannot schedule makes rocprofv3 abort and then hang).
#   tools/pmc_passes.sh <out_prefix> [mem|sq|all|mfma] -- <program> [args...]
out=$1; which=$2; shift; shift; shift
mem=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_HIT_sum TCC_MISS_sum"
 "TCC_REQ_sum TCP_TCC_READ_REQ_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
)
sq=(
 "SQ_WAVES SQ_BUSY_CYCLES"
 "SQ_WAVE_CYCLES SQ_LEVEL_WAVES"
 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
 "SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
 "TA_BUSY_avr GRBM_GUI_ACTIVE"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "MeanOccupancyPerActiveCU"
)
mfma=(
 "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
 "MfmaUtil"
 "GRBM_GUI_ACTIVE"
)
sets=()
[ "$which" = mem ] || [ "$which" = all ] && sets+=("${mem[@]}")
[ "$which" = sq ] || [ "$which" = all ] && sets+=("${sq[@]}")
[ "$which" = mfma ] && sets+=("${mfma[@]}")
i=0
for set in "${sets[@]}"; do
  d=${out}_$i
  timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- "$@" > $d.log 2>&1 || echo "pass $i ($set) failed rc=$?"
  i=$((i+1))
done
