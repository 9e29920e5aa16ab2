#!/bin/bash
# kernel-trace statistics of a short bench run (no PMC): tools/ktrace.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kt_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --no-cpu-baseline --no-e2e --steps 4 --warmup 1 "$@" > $out/bench.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/t_kernel_stats.csv")))
for r in rows:
    n=r['Name']
    if n.startswith('void k_') or n.startswith('k_') or 'rocprim' in n:
        print(f"{n[:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
