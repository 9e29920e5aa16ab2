#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "multiword or repeat_rich" 2>&1 | tail -2
python bench.py --kmer-size 63 --workload c2_repeats_10Mx150 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k63 c2 repeats', round(b['ms_per_step'],3), b['engine_stats'], b['stage_ms'])"
for i in 1 2 3; do bash tools/ab.sh "default dsk_amd/variants/libdskgpu_xnackoff.so" --no-repeat-rich --no-human-standin --no-place-compare; done
