#!/bin/bash
# r05 step 22: k_count_valid for 32 < k <= 64 as a 96-base smear, four words per trip -- parity, then the sample1 stage at k = 63
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s22; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -5 > $out/parity.log
cat $out/parity.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_invariants and 63" 2>&1 | tail -3 >> $out/parity.log
tail -3 $out/parity.log
for r in 1 2; do
  python3 bench.py --kmer-size 63 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
done > $out/k63.log 2>&1
cat $out/k63.log
