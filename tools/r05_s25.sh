#!/bin/bash
# r05 step 25: shader clock and power while the k = 63 steps run (are the "two speeds" of the two-word scatters a clock / power state of the box?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s25; mkdir -p $out
rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40 > $out/idle.txt
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Power (W)\|Average Graphics\|Current Socket" | tr '\n' ' '; echo; sleep 0.05; done ) > $out/samples.txt 2>&1 &
smi=$!
for ks in 63 31; do
  python3 bench.py --kmer-size $ks --steps 300 --warmup 2 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('k', $ks, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if k in ('scatter1','scatter2','count','sort')})"
done > $out/bench.txt 2>&1
kill $smi 2>/dev/null
cat $out/bench.txt
head -12 $out/idle.txt
sort $out/samples.txt | uniq -c | sort -rn | head -25
