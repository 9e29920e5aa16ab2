#!/bin/bash
out=gpurun_out/r04j; mkdir -p $out
for i in 1 2 3; do
  for lib in dsk_amd/variants/libdskgpu_r03.so dsk_amd/libdskgpu.so; do
    echo "== $lib"; DSKGPU_LIB=$PWD/$lib python3 tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 2>&1 | grep "scatter side\|stages"
  done
done
python -m pytest tests/test_gpu_parity.py -x -q -k "repeat_rich or receive_side or region_chains or poly_a or mostly_invalid or row_sort or multiword" > $out/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_sel.log | cut -c1-300
python bench.py --kmer-size 63 --workload c2_repeats_10Mx150 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k63 c2 repeats', round(b['ms_per_step'],3), b['engine_stats'], b['stage_ms'])"
