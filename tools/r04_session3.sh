#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "row_sort" > $out/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_sel.log
DSKGPU_VERBOSE=1 timeout 900 python tools/human_standin.py 75 31 2 > $out/human_shard.log 2>&1; echo "human shard rc $?"; grep "row sort\|^{" $out/human_shard.log | tail -8
DSKGPU_VERBOSE=1 timeout 1500 python tools/human_standin.py 600 31 2 > $out/human_full.log 2>&1; echo "human full rc $?"; grep "level 0\|row sort\|overflow\|exact path\|^{\|Error" $out/human_full.log | tail -30
