#!/bin/bash
# r06 step 14: staging copy fixed (T * per >= n), device parse through the binary, three-word keys (65 <= k <= 96) against the four-word path
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s14; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_raw_parse.py -x -q 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_cli_gpu.py -x -q -k "device_parse or messy or simple_test or engine_is" --durations=5 2>&1 | tail -12 > $o/cli.log; cat $o/cli.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "push_reads" 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py -x -q -k "four_word or four_words or span_borders or large_k" --durations=5 2>&1 | tail -15 > $o/w3.log; cat $o/w3.log
STRESS_KS=65,66,72,80,89,95,96 timeout 1200 python3 tools/stress_random.py 13000 80 > $o/three_word_seeds_13000.log 2>&1; tail -2 $o/three_word_seeds_13000.log | cut -c1-250
for k in 80 96; do
  python3 bench.py --kmer-size $k --steps 10 --warmup 2 --no-e2e --no-human-standin --no-repeat-rich --no-place-compare 2>/dev/null | grep '^{"metric"' > $o/bench_k${k}_w3.json
  DSKGPU_FORCE_W4=1 python3 bench.py --kmer-size $k --steps 10 --warmup 2 --no-e2e --no-human-standin --no-repeat-rich --no-place-compare 2>/dev/null | grep '^{"metric"' > $o/bench_k${k}_w4.json
  python3 -c "
import json
for w in ('w3','w4'):
    d=json.loads(open('$o/bench_k${k}_'+w+'.json').read())
    print('k=$k', w, d['ms_per_step'], d.get('stage_ms'))
"
done
