#!/bin/bash
# big buffers through the virtual-memory API (DSKGPU_VMM, default on) against hipMalloc, with and without the 8-candidate placement:
# several processes each (the classes change from process to process)
run() { # label, env, args
  python bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-repeat-rich --no-k63 --steps 10 --warmup 3 $3 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=b['stage_ms']; print('$1', round(b['ms_per_step'],3), 'first', b['first_step_s'], 'L1', s['scatter1'], 'L2', s['scatter2'], 'count', s['count'], 'sort', s['sort'], 'enc', s['encode'])"
}
for i in 1 2 3 4; do
  DSKGPU_VMM=1 run "k31 vmm      " x "--no-place"
  DSKGPU_VMM=0 run "k31 malloc   " x "--no-place"
  DSKGPU_VMM=0 run "k31 placed   " x ""
done
for i in 1 2 3; do
  DSKGPU_VMM=1 run "k63 vmm      " x "--no-place --kmer-size 63"
  DSKGPU_VMM=0 run "k63 malloc   " x "--no-place --kmer-size 63"
  DSKGPU_VMM=0 run "k63 placed   " x "--kmer-size 63"
done
