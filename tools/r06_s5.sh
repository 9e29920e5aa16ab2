#!/bin/bash
# r06 step 5: the packed count table (key | count in one LDS word): micro benchmark first, then A/B in the real step, then the placement counters (r06_s4)
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s5; mkdir -p $o
timeout 120 tools/micro/count_tag > $o/count_tag.txt 2>&1; cat $o/count_tag.txt
for rep in 1 2; do
  for v in 0 1; do
    echo "== DSKGPU_NO_PACKED=$v"; DSKGPU_NO_PACKED=$v python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-repeat-rich --no-k63 --no-place-compare --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items() if k in ('scatter1','scatter2','count','sort')}, d['n_distinct'], d['n_solid'])"
  done
done 2>&1 | tee $o/ab.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_invariants and c2_10Mx150-31 or poly_a or overflow_retry or partition_order or full_size_repeat" 2>&1 | tail -5 | tee $o/tests.txt
bash tools/r06_s4.sh > $o/s4.txt 2>&1; tail -60 $o/s4.txt | cut -c1-300
