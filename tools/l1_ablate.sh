#!/bin/bash
# where the level-1 scatter spends its time: an ablated launch (DSK_L1_ABLATE build: dsk_amd/libdskgpu_abl.so) before the real one in
# every step; abl 1 = no global stores, 2 = no write-out, 3 = no staging either, 4 = generation + mixer only.   tools/l1_ablate.sh
for a in 1 2 3 4; do
  DSKGPU_L1_ABL=$a DSKGPU_LIB=$PWD/dsk_amd/libdskgpu_abl.so python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-repeat-rich 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('abl $a: ablated', round(s.get('scatter1_abl',-1),3), 'complete', round(s['scatter1'],3), 'scatter2', round(s['scatter2'],3))"
done
