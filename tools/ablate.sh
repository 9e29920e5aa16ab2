#!/bin/bash
# usage: tools/ablate.sh "<DBG values>" [bench args]
vals=$1; shift
for v in $vals; do
  echo "== DSKGPU_DBG=$v"
  DSKGPU_DBG=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['stage_ms'])"
done
