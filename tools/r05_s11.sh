#!/bin/bash
# r05 step 11: one read-back per stage (k_gather_back / k_sort_back) -- parity subset, then A/B against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s11
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -8 > gpurun_out/r05_s11/parity.log
cat gpurun_out/r05_s11/parity.log
bash tools/ab_rep.sh "dsk_amd/libdskgpu_base.so default" 3 > gpurun_out/r05_s11/ab.log 2>&1
cat gpurun_out/r05_s11/ab.log
