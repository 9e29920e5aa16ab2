#!/bin/bash
# r05 step 26: passes of a multi-pass job compact their rows straight into the job's accumulators -- parity (multi-pass tests, 120 random seeds), stand-in time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s26; mkdir -p $out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > $out/suite.log
cat $out/suite.log
timeout 900 python3 tools/stress_random.py 9000 60 > $out/stress2.log 2>&1; tail -1 $out/stress2.log
STRESS_KS=15,21,27,31,32 timeout 900 python3 tools/stress_random.py 9100 60 > $out/stress1.log 2>&1; tail -1 $out/stress1.log
grep -c "passes [2-9]" $out/stress1.log $out/stress2.log
python3 tools/human_standin.py 600 31 1 2>&1 | grep '^{"workload"' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('count_s','n_passes','n_read_sweeps','hbm_used_gb','kmer_occurrences_per_s')}); print(d['stage_ms'])" | tee $out/human.log
