cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s19; mkdir -p $o
( time python3 bench.py > $o/r06c_bench.json 2> $o/r06c_bench.err ) 2> $o/r06c_bench.time; tail -3 $o/r06c_bench.time
bash tools/pmc_any.sh r06c_mg_fetch "FETCH_SIZE" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_fetch.txt 2>&1
bash tools/pmc_any.sh r06c_mg_write "WRITE_SIZE" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_write.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"
SQ2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
bash tools/pmc_any.sh r06c_mg_sq1 "$SQ1" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_sq1.txt 2>&1
bash tools/pmc_any.sh r06c_mg_sq2 "$SQ2" tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > $o/mg_sq2.txt 2>&1
rm -rf gpurun_out/pmc_r06c_*/*.csv
for f in mg_fetch mg_write mg_sq1 mg_sq2; do echo "== $f"; grep "k_sk_\|k_scatter" $o/$f.txt | cut -c1-400; done
