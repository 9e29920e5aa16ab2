#!/bin/bash
# k = 63 with sub-partitions of 2560 keys (regions of 4092): parity, bench; and the count kernel's time against the sub-partition size at k = 31
python -m pytest tests/test_gpu_parity.py -x -q -k "two_word or region_chains or repeat_rich or multiword or table_overflow or multi_pass_over or receive_side or four_word" 2>&1 | tail -3
python bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 8 --warmup 3 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k63 c2', round(b['ms_per_step'],3), b['engine_stats'], b['stage_ms']); print('   repeats', b['repeat_rich']['ms_per_step'], b['repeat_rich']['stage_ms'])"
DSKGPU_COUNT_MW_V1=1 python bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 8 --warmup 3 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k63 c2 index-table kernel', round(b['ms_per_step'],3), b['engine_stats'], b['stage_ms'])"
DSKGPU_TABLE_MAXLOAD=700 python bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k31 c2 maxload 700 (retry with sub-partitions of 1450 keys; stages summed over both attempts)', round(b['ms_per_step'],3), b['engine_stats'], b['stage_ms'])"
