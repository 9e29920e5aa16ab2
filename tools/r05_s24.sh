#!/bin/bash
# r05 step 24: what the placement probe sees at k = 63 (22 / 28 GB buffers), and whether more candidates change the class
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s24; mkdir -p $out
one() { python3 bench.py --kmer-size 63 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich 2> $out/err.txt | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if k in ('scatter1','scatter2','count','sort')})"; grep "placement of" $out/err.txt | grep -v "0\.[0-9]* GB"; }
for v in "" "DSKGPU_PLACE=16" "DSKGPU_PLACE=2" "DSKGPU_PLACE=1"; do
  echo "== ${v:-default (8 candidates)}"
  env DSKGPU_VERBOSE=1 $v bash -c "$(declare -f one); out=$out; one"
done > $out/log.txt 2>&1
cat $out/log.txt
