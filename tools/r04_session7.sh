#!/bin/bash
out=gpurun_out/r04h; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "receive_side or multi_pass" > $out/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -15 $out/pytest_sel.log | cut -c1-400
