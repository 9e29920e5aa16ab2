#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args]   -> gpurun_out/pmc_<tag>/
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p -- python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --steps 2 --warmup 1 "$@" > $out/bench.log 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$out/p_counter_collection.csv")))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name']
    if k.startswith('void k_') or k.startswith('k_'):
        acc[k[:44]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
