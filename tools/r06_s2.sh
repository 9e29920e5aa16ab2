#!/bin/bash
# r06 step 2: the sender kernels with compile-time k / m and the trimmed record phase: parity of everything that goes through records,
# then the stage times of one emulated rank of 8 (k = 31, 63; the 25 M-read shard) and of the human stand-in
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s2; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sender or super_kmer or repartition or multi_pass or group or sliced or minimizers or records or standin_small or exchange" > $o/tests.txt 2>&1
tail -5 $o/tests.txt
for a in "8 31" "8 63" "8 31 0 c3_shard_25Mx150" "8 31 0 c2_10Mx150 4"; do
  echo "== mg_stage_times $a"; python3 tools/mg_stage_times.py $a 2>&1 | grep -v amdgpu.ids | tee -a $o/mg.txt
  echo "== generic"; DSKGPU_SK_GENERIC=1 python3 tools/mg_stage_times.py $a 2>&1 | grep "scatter side\|stages" | tee -a $o/mg.txt
done
python3 tools/human_standin.py 600 31 1 > $o/hs.txt 2>&1; tail -1 $o/hs.txt | cut -c1-1500
