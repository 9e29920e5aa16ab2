#!/bin/bash
# r06 step 13: the dsk binary with -device-parse 1 (text parsed on the GPU), the staging copy on four threads, N = 4 on a shared GPU
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s13; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_raw_parse.py -x -q 2>&1 | tail -5 > $o/raw.log; cat $o/raw.log
timeout 1800 python -m pytest tests/test_cli_gpu.py -x -q -k "device_parse or messy or simple_test or engine_is or push" --durations=5 2>&1 | tail -25 > $o/cli.log; cat $o/cli.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "push_reads" 2>&1 | tail -3
python3 - > $o/e2e.json 2> $o/e2e.err <<'PY'
import json, bench
print(json.dumps(bench.e2e_block(31, 2, 400.0)))
PY
python3 -c "
import json
d=json.load(open('$o/e2e.json'))
for w in ('ecoli50x','c2_10Mx150'):
    for leg,v in d[w].items():
        if isinstance(v,dict) and 'wall_s' in v: print(w, leg, {k:v.get(k) for k in ('ingest_s','count_s','write_s','total_s','wall_s','banks_parsed_on_device')})
"; tail -3 $o/e2e.err
DSK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 4 --workload c2_10Mx150 --check-parity --steps 2 --warmup 1 2>$o/bench_n4.err | grep '^{"metric"' > $o/bench_n4_shared_gpu_development.json; cut -c1-300 $o/bench_n4_shared_gpu_development.json; grep -v Gloo $o/bench_n4.err | tail -3
