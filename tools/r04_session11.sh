#!/bin/bash
# k = 63: the top-word count table (k_count2v3) -- parity, the forced fallback, and the bench line beside the index-table kernel
python -m pytest tests/test_gpu_parity.py -x -q -k "two_word or region_chains or repeat_rich or multiword or full_size_invariants" 2>&1 | tail -3
for v in 0 1; do
DSKGPU_COUNT_MW_V1=$v python bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 8 --warmup 3 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k63 c2 mw_v1=$v', round(b['ms_per_step'],3), b['engine_stats'], b['stage_ms'])"
done
