#!/bin/bash
# r06 step 10: partition sort with 1024-thread blocks (A/B), the emulated rank of 8 with its rows in partition order, the multi-pass tests again
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s10; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "partition_order or multi_pass" 2>&1 | tail -3 | cut -c1-200 | tee $o/tests.txt
for rep in 1 2; do bash tools/ab.sh "default dsk_amd/variants/libdskgpu_ps1024.so" --no-human-standin --no-repeat-rich --no-place-compare 2>&1 | cut -c1-400; done | tee $o/ab.txt
for a in "8 31 0 c2_10Mx150 0 partition" "8 31 0 c3_shard_25Mx150 0 partition" "8 63 0 c2_10Mx150 0 partition"; do echo "== $a"; python3 tools/mg_stage_times.py $a 2>&1 | grep "scatter side\|stages" | tee -a $o/mg.txt; done
