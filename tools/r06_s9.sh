#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s9; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "partition_order or multi_pass or human_standin or row_sort" 2>&1 | tail -5 | cut -c1-300 | tee $o/tests.txt
