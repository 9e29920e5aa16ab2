#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_list
rocprofv3 -L > gpurun_out/pmc_list/counters.txt 2>&1
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCC_BUSY_avr TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum" "TCC_TAG_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  bash tools/pmc.sh r2n_$i "$set" > gpurun_out/pmc_list/set_$i.txt 2>&1
  echo "== $set"; grep "k_scatter<1, 0, 1\|k_scatter_al<1, 2\|k_count1" gpurun_out/pmc_list/set_$i.txt | cut -c1-400
done
