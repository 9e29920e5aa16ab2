#!/bin/bash
# round 5, session 2: the 32-bit-tag count table (micro benchmark) + SQ counters of the three big kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s2
./tools/micro/count_tag > gpurun_out/r05_s2/count_tag.txt 2>&1
cat gpurun_out/r05_s2/count_tag.txt
bash tools/pmc.sh r05_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" --no-repeat-rich > gpurun_out/r05_s2/sq1.txt 2>&1
bash tools/pmc.sh r05_sq2 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" --no-repeat-rich > gpurun_out/r05_s2/sq2.txt 2>&1
tail -30 gpurun_out/r05_s2/sq1.txt; tail -30 gpurun_out/r05_s2/sq2.txt
