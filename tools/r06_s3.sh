#!/bin/bash
# r06 step 3: the partition-order row sort (partsort.h): parity, then the bench line with both row orders; the tests that did not run in the suite
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s3; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "partition_order or row_sort or sender_with_compile" > $o/tests.txt 2>&1
tail -15 $o/tests.txt
python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-repeat-rich --steps 20 --warmup 3 > $o/bench.json 2> $o/bench.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_s3/bench.json") if l.startswith('{"metric"')][-1])
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_no_place", "ms_per_step_global_order", "sort_ms_other_row_order", "row_order", "first_step_s")})
print(d["stage_ms"] if "stage_ms" in d else "")
print(d["roofline"].get("partition_plus_hash"), d["roofline"].get("step_frac"))
print("k63", d.get("k63", {}).get("ms_per_step"), d.get("k63", {}).get("stage_ms"))
PY
