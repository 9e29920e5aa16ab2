#!/bin/bash
# r06 step 1: counters of the RECORD path before it is rebuilt: FETCH_SIZE / WRITE_SIZE and the two SQ sets
# of profiles/r05_sq.md for k_sk_scatter<true> (8 owners, c2 shard), k_scatter<1,2,..> (level 1 from records) and, on the human
# stand-in (60 virtual owners), k_sk_scatter<true> / level 1 from records.
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s1; mkdir -p $o
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
SQ2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
python3 tools/mg_stage_times.py 8 31 > $o/mg_plain.txt 2>&1
bash tools/pmc_any.sh r06_mg_fetch "FETCH_SIZE" tools/mg_stage_times.py 8 31 > $o/mg_fetch.txt 2>&1
bash tools/pmc_any.sh r06_mg_write "WRITE_SIZE" tools/mg_stage_times.py 8 31 > $o/mg_write.txt 2>&1
bash tools/pmc_any.sh r06_mg_sq1 "$SQ1" tools/mg_stage_times.py 8 31 > $o/mg_sq1.txt 2>&1
bash tools/pmc_any.sh r06_mg_sq2 "$SQ2" tools/mg_stage_times.py 8 31 > $o/mg_sq2.txt 2>&1
python3 tools/human_standin.py 600 31 1 > $o/hs_plain.txt 2>&1
bash tools/pmc_any.sh r06_hs_fetch "FETCH_SIZE" tools/human_standin.py 600 31 0 > $o/hs_fetch.txt 2>&1
bash tools/pmc_any.sh r06_hs_write "WRITE_SIZE" tools/human_standin.py 600 31 0 > $o/hs_write.txt 2>&1
bash tools/pmc_any.sh r06_hs_sq1 "$SQ1" tools/human_standin.py 600 31 0 > $o/hs_sq1.txt 2>&1
bash tools/pmc_any.sh r06_hs_sq2 "$SQ2" tools/human_standin.py 600 31 0 > $o/hs_sq2.txt 2>&1
rm -rf gpurun_out/pmc_r06_*/*.csv gpurun_out/pmc_r06_*/*/*.csv
for f in $o/*.txt; do echo "== $f"; grep -v "^$" $f | tail -25 | cut -c1-600; done
