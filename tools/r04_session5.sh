#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "multi_pass or row_sort or four_word_multi" > $out/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_sel.log
DSKGPU_VERBOSE=1 timeout 900 python tools/check_invariants.py c3_200Mx150 31 > $out/c3_200M.log 2>&1; echo "c3_200M rc $?"; grep -v "^\[dskgpu\]   \|^\[dskgpu\] pass\|find_heavy\|level 2:" $out/c3_200M.log | tail -8
DSKGPU_VERBOSE=1 timeout 900 python tools/check_invariants.py c3_200Mx150 63 > $out/c3_200M_k63.log 2>&1; echo "c3_200M k63 rc $?"; grep -v "^\[dskgpu\]   \|^\[dskgpu\] pass\|find_heavy\|level 2:" $out/c3_200M_k63.log | tail -8
DSKGPU_VERBOSE=1 timeout 1500 python tools/human_standin.py 600 31 2 > $out/human_full.log 2>&1; echo "human full rc $?"; grep "level 0\|row sort\|overflow\|exact path\|^{\|Error" $out/human_full.log | tail -30
