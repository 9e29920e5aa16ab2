#!/bin/bash
# round 5, session 5: where the dsk binary's wall clock goes (c2 and E. coli), with and without the engine teardown
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s5
for wl in c2_10Mx150 ecoli50x; do
  echo "== $wl, no teardown (product)"; python tools/e2e_phase2.py $wl 4 2>&1 | grep -v amdgpu.ids
  echo "== $wl, DSK_TEARDOWN=1"; DSK_TEARDOWN=1 python tools/e2e_phase2.py $wl 3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_s5/phases.txt 2>&1
cat gpurun_out/r05_s5/phases.txt
