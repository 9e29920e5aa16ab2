#!/bin/bash
# round 5, session 7: chunk size of the bank -> engine hand-over (dsk binary, c2 FASTQ)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s7
for mb in 8 16 32 64 16 8 32 64; do
  echo "== chunk $mb MB"; DSK_CHUNK_MB=$mb python tools/e2e_phase2.py c2_10Mx150 4 2>&1 | grep "^wall" | tail -3
done > gpurun_out/r05_s7/chunks.txt 2>&1
cat gpurun_out/r05_s7/chunks.txt
