#!/bin/bash
# where the level-1 scatter spends its time: a build with timing scaffolding (tools/micro/l1_ablate.patch applied to a scratch copy of
# dsk_amd/csrc: the product sources carry none of it) launches a cut-down copy of the kernel before the real one in every step;
# abl 1 = no global stores, 2 = no write-out, 3 = no staging either, 4 = generation + mixer only.   tools/l1_ablate.sh
set -e
tmp=$(mktemp -d); cp -r dsk_amd/csrc "$tmp/csrc"; mkdir -p "$tmp/include"; cp include/dskgpu.h "$tmp/include/"
(cd "$tmp/csrc" && patch -p0 < "$OLDPWD/tools/micro/l1_ablate.patch" && sed -i 's#../../include/dskgpu.h#../include/dskgpu.h#' dskgpu.hip group.hip && \
 /opt/rocm/bin/hipcc -DDSK_L1_ABLATE -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -shared -o "$OLDPWD/dsk_amd/libdskgpu_abl.so" dskgpu.hip group.hip -ldl)
for a in 1 2 3 4; do
  DSKGPU_L1_ABL=$a DSKGPU_LIB=$PWD/dsk_amd/libdskgpu_abl.so python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-repeat-rich --no-human-standin 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('abl $a: ablated', round(s.get('scatter1_abl',-1),3), 'complete', round(s['scatter1'],3), 'scatter2', round(s['scatter2'],3))"
done
