#!/bin/bash
# r06 step 11: the whole suite at the round's kernels, then the profile set (tools/profile_set.sh r06a), then the SQ counters of the rebuilt
# record path and of the step's kernels (same two sets as profiles/r05_sq.md / r06_records.md)
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s11; mkdir -p $o
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=6 2>&1 | tail -14 > $o/suite.log; cat $o/suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_set.sh r06a > $o/profile_set.log 2>&1; tail -3 $o/profile_set.log | cut -c1-300
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
SQ2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
bash tools/pmc_any.sh r06b_mg_fetch "FETCH_SIZE" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_fetch.txt 2>&1
bash tools/pmc_any.sh r06b_mg_write "WRITE_SIZE" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_write.txt 2>&1
bash tools/pmc_any.sh r06b_mg_sq1 "$SQ1" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_sq1.txt 2>&1
bash tools/pmc_any.sh r06b_mg_sq2 "$SQ2" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_sq2.txt 2>&1
bash tools/pmc.sh r06b_sq1 "$SQ1" --no-repeat-rich > $o/bench_sq1.txt 2>&1
bash tools/pmc.sh r06b_sq2 "$SQ2" --no-repeat-rich > $o/bench_sq2.txt 2>&1
rm -rf gpurun_out/pmc_r06b_*/*.csv
for f in mg_fetch mg_write mg_sq1 mg_sq2 bench_sq1 bench_sq2; do echo "== $f"; grep "k_sk_\|k_scatter\|k_part\|k_count\|k_encode" $o/$f.txt | cut -c1-500; done
