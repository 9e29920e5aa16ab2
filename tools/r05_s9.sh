#!/bin/bash
# round 5, session 9: the row sort's first step from the count kernel's regions (no k_compact pass) -- parity, then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s9
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_sort or golden or synthetic or region_chains or repeat_rich_reads or randomized or poly_a or histogram_free or multi_gpu_path or group_count or receive_side or abundance_window or empty or full_size_repeat_rich" 2>&1 | tail -8
for r in 1 2 3; do
  for arm in "DSKGPU_SORT_COMPACT=1" "X=1"; do
    echo "== $arm"; env $arm python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-human-standin --no-k63 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if k in ('count','compact','sort','scan_solid')}, 'rr', d['repeat_rich']['ms_per_step'], {k:round(v,3) for k,v in d['repeat_rich']['stage_ms'].items() if k in ('compact','sort')})"
  done
done > gpurun_out/r05_s9/ab.txt 2>&1
cat gpurun_out/r05_s9/ab.txt
