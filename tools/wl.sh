#!/bin/bash
# stage times + stats of one bench workload:  tools/wl.sh "<workloads>" [bench args]
wls=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in $wls; do
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --workload $wl "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'],3), int(d['n_kmers']), int(d['n_distinct']), {k:round(v,3) for k,v in d['stage_ms'].items()}, d.get('engine_stats'))
except Exception as e: print('$wl FAILED', e)"
done
