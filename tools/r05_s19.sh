#!/bin/bash
# r05 step 19: k_encode four bytes at a time (two gather multiplies, a 32-bit base mask) -- encode / enumerate parity, then A/B (encode stage shown)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s19; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -5 > $out/parity.log
cat $out/parity.log
for r in 1 2 3; do for lib in dsk_amd/libdskgpu_prev.so dsk_amd/libdskgpu.so; do
  echo -n "$lib  "
  DSKGPU_LIB=$PWD/$lib python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --no-repeat-rich 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if k in ('encode','sample1','scatter1','scatter2','count','sort')})"
done; done > $out/ab.log 2>&1
cat $out/ab.log
