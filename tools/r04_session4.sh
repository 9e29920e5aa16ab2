#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "receive_side or row_sort or group or exchange or multi_gpu or sliced or step_in_slices or super_kmer or repartition" > $out/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_sel.log
