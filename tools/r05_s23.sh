#!/bin/bash
# r05 step 23: the two speeds of the two-word scatters -- does a pause between processes (the driver still releasing the last process's
# 50 GB?) decide the class?  six k = 63 runs back to back, six with 8 s of idle time before each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s23; mkdir -p $out
one() { python3 bench.py --kmer-size 63 --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if k in ('scatter1','scatter2','count','sort')}, 'first_step_s', d.get('first_step_s'))"; }
echo "back to back" > $out/log.txt
for r in 1 2 3 4 5 6; do one >> $out/log.txt 2>&1; done
echo "8 s pause before each" >> $out/log.txt
for r in 1 2 3 4 5 6; do sleep 8; one >> $out/log.txt 2>&1; done
cat $out/log.txt
