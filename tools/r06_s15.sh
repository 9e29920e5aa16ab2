#!/bin/bash
# r06 step 15: device parse through the binary after the fixes, the staging remainder test, the e2e block with the device-parse legs
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s15; mkdir -p $o
timeout 1800 python -m pytest tests/test_cli_gpu.py tests/test_gpu_raw_parse.py -x -q -k "device_parse or messy or simple_test or engine_is or raw_parse" --durations=5 2>&1 | tail -12 > $o/cli.log; cat $o/cli.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "push_reads or four_word" 2>&1 | tail -3
python3 - > $o/e2e.json 2> $o/e2e.err <<'PY'
import json, bench
print(json.dumps(bench.e2e_block(31, 2, 400.0)))
PY
python3 -c "
import json
d=json.load(open('$o/e2e.json'))
for w in ('ecoli50x','c2_10Mx150'):
    for leg,v in d[w].items():
        if isinstance(v,dict) and 'wall_s' in v: print(w, leg, {k:v.get(k) for k in ('ingest_s','count_s','write_s','total_s','wall_s','banks_parsed_on_device','kmers_nb_valid')})
"; tail -3 $o/e2e.err
