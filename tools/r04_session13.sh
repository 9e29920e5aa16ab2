#!/bin/bash
# k = 63: level 1 from the reads with wave-uniform window offsets and the next tile's words prefetched
python -m pytest tests/test_gpu_parity.py -x -q -k "two_word or region_chains or repeat_rich or enumerate_two or multi_pass_over or mostly_invalid or golden_T or iupac or edge" 2>&1 | tail -3
python bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 8 --warmup 3 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k63 c2', round(b['ms_per_step'],3), b['engine_stats'], b['stage_ms']); print('   repeats', b['repeat_rich']['ms_per_step'], b['repeat_rich']['stage_ms'])"
