#!/bin/bash
out=gpurun_out/r04i; mkdir -p $out
python tools/e2e.py ecoli50x 31 > $out/e2e.log 2>&1; python tools/e2e_phase.py ecoli50x >> $out/e2e.log 2>&1; tail -8 $out/e2e.log
python bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-human-standin --steps 10 --warmup 3 > $out/bench_k63.json 2>$out/bench_k63.err; python - <<PY
import json
b=json.loads(open("$out/bench_k63.json").read().strip().splitlines()[-1])
print("k63 c2", round(b["ms_per_step"],3), b.get("ms_per_step_no_place"), b["stage_ms"])
PY
python bench.py --kmer-size 63 --workload c2_repeats_10Mx150 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 5 --warmup 2 > $out/bench_k63_rep.json 2>$out/bench_k63_rep.err; python - <<PY
import json
b=json.loads(open("$out/bench_k63_rep.json").read().strip().splitlines()[-1])
print("k63 c2 repeats", round(b["ms_per_step"],3), b["engine_stats"], b["stage_ms"])
PY
python bench.py --workload c3_shard_25Mx150 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-repeat-rich --steps 5 --warmup 2 > $out/bench_shard.json 2>$out/bench_shard.err; python - <<PY
import json
b=json.loads(open("$out/bench_shard.json").read().strip().splitlines()[-1])
print("shard 25M", round(b["ms_per_step"],3), b["engine_stats"], b["stage_ms"])
PY
