#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in ecoli50x c2_10Mx150 c3_shard_25Mx150; do
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'],3), int(d['n_kmers']), {k:round(v,3) for k,v in d['stage_ms'].items()})"
done
