#!/bin/bash
# A/B of several builds of libdskgpu.so in one GPU-box session (timing only):
#   tools/ab.sh "<lib1> <lib2> ..." [bench args]       (lib = path, or "default"; "ENV=val:lib" sets an env var for that arm)
libs=$1; shift
for spec in $libs; do
  envs=""; lib=$spec
  if [[ $spec == *:* ]]; then envs=${spec%%:*}; lib=${spec#*:}; fi
  [ "$lib" = default ] && lib=dsk_amd/libdskgpu.so
  echo "== $spec"
  env $envs DSKGPU_LIB=$PWD/$lib python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()}, int(d['n_distinct']), int(d['n_solid']))
except Exception as e: print('FAILED', e)"
done
