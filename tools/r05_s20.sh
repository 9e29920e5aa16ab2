#!/bin/bash
# r05 step 20: the two-word row sort reads the count kernel's regions (no k_compact<2>) -- the whole GPU suite, then A/B at k = 63
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s20; mkdir -p $out
timeout 2700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > $out/suite.log
cat $out/suite.log
for r in 1 2 3; do for lib in dsk_amd/libdskgpu_prev.so dsk_amd/libdskgpu.so; do
  echo -n "$lib  "
  DSKGPU_LIB=$PWD/$lib python3 bench.py --kmer-size 63 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if k in ('scatter1','scatter2','count','compact','sort')})"
done; done > $out/ab63.log 2>&1
cat $out/ab63.log
