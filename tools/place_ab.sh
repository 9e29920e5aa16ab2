#!/bin/bash
# bench stage times of separate processes, without and with DSKGPU_PLACE: tools/place_ab.sh [K=8] [runs=4]
K=${1:-8}; runs=${2:-4}
for i in $(seq 1 $runs); do
  for k in 0 $K; do
    DSKGPU_PLACE=$k python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-repeat-rich 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('place $k:', round(d['ms_per_step'],2), 'scatter1', round(s['scatter1'],2), 'scatter2', round(s['scatter2'],2), 'count', round(s['count'],2), 'sort', round(s['sort'],2))"
  done
done
