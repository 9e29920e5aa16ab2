#!/bin/bash
# r05 step 21: SQ counters of the step's kernels at k = 63 (two-word keys), the same two passes as profiles/r05_sq.md at k = 31
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s21
bash tools/pmc.sh r05_sq63a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" --no-repeat-rich --kmer-size 63 > gpurun_out/r05_s21/sq1.txt 2>&1
bash tools/pmc.sh r05_sq63b "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" --no-repeat-rich --kmer-size 63 > gpurun_out/r05_s21/sq2.txt 2>&1
tail -30 gpurun_out/r05_s21/sq1.txt; tail -30 gpurun_out/r05_s21/sq2.txt
rm -rf gpurun_out/pmc_r05_sq63a/*.csv gpurun_out/pmc_r05_sq63b/*.csv
