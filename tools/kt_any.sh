#!/bin/bash
# kernel-trace statistics of any python tool: tools/kt_any.sh <tag> <script> [args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kt_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 "$@" > $out/run.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/t_kernel_stats.csv")))
for r in rows[:28]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
tail -2 $out/run.log | cut -c1-300
