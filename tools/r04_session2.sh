#!/bin/bash
# round-4 GPU session 2: multi-pass redesign (small passes, 16-pass level 0 with sampled slices, group-wise row sort) -- targeted tests, then sizes
out=gpurun_out/r04c; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "multi_pass or row_sort or failed_slice_gate or four_word_multi or region_chains or repeat_rich or poly_a" > $out/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_sel.log
DSKGPU_VERBOSE=1 timeout 900 python tools/check_invariants.py c3_200Mx150 31 > $out/c3_200M.log 2>&1; echo "c3_200M rc $?"; grep -v "^\[dskgpu\]   " $out/c3_200M.log | tail -12
DSKGPU_VERBOSE=1 timeout 900 python tools/human_standin.py 75 31 1 > $out/human_shard.log 2>&1; echo "human shard rc $?"; grep -v "^\[dskgpu\]   " $out/human_shard.log | tail -8
DSKGPU_VERBOSE=1 timeout 1500 python tools/human_standin.py 600 31 1 > $out/human_full.log 2>&1; echo "human full rc $?"; grep -v "^\[dskgpu\]   " $out/human_full.log | tail -30
