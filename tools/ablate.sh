#!/bin/bash
# usage: tools/ablate.sh <1|2> "<DBG values>" [bench args]   (level-1 or key-array scatter ablations; timing only)
lvl=$1; vals=$2; shift 2
for v in $vals; do
  echo "== DSKGPU_DBG$lvl=$v"
  env DSKGPU_DBG$lvl=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:v for k,v in d['stage_ms'].items() if k in ('hist1','scatter1','hist2','scatter2','count')})"
done
