#!/bin/bash
# chosen probe time of the two big buffers against the scatter kernels' times, over several processes: tools/place_corr.sh [runs]
for i in $(seq 1 ${1:-6}); do
  DSKGPU_VERBOSE=1 python3 bench.py --no-cpu-baseline --no-e2e --no-repeat-rich --steps 6 --warmup 2 2>&1 | grep -E "placement of 1[0-9]|metric" | python3 -c "
import sys,json,re
pr=[]
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stage_ms']; print('probes', pr, 'scatter1', round(s['scatter1'],2), 'scatter2', round(s['scatter2'],2), 'step', round(d['ms_per_step'],2))
    else:
        m=re.search(r'of ([0-9.]+) GB.*probe ms (.*)', l); vals=m.group(2).split(); best=[v for v in vals if v.endswith('*')][0]
        pr.append((m.group(1), best, min(float(v.strip('*')) for v in vals), max(float(v.strip('*')) for v in vals)))"
done
