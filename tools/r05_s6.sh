#!/bin/bash
# round 5, session 6: dsk binary phases after the parser / push changes; push + CLI parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s6
timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "push or simple_test or known_answer or nb_gpus_writes or fewer_reads or span_borders" 2>&1 | tail -5
for wl in c2_10Mx150 ecoli50x; do
  echo "== $wl"; python tools/e2e_phase2.py $wl 6 2>&1 | grep -v "amdgpu.ids\|\[dsk\]"
done > gpurun_out/r05_s6/phases.txt 2>&1
cat gpurun_out/r05_s6/phases.txt
