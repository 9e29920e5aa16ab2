#!/bin/bash
# (the DSKGPU_PLACE_TRIALS patch this script measured was reverted: profiles/r05_place_trials.txt)
# r05 step 16: placement by trial counts (DSKGPU_PLACE_TRIALS, default 6) against the probe alone (0): separate processes, k = 31 and 63
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s16; mkdir -p $out
DSKGPU_VERBOSE=1 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich 2> $out/verbose31.err | tail -1 | cut -c1-400
grep "placement" $out/verbose31.err | tail -20
bash tools/ab_rep.sh "DSKGPU_PLACE_TRIALS=0:default default" 4 > $out/ab31.log 2>&1
cat $out/ab31.log
bash tools/ab_rep.sh "DSKGPU_PLACE_TRIALS=0:default default" 4 --kmer-size 63 > $out/ab63.log 2>&1
cat $out/ab63.log
