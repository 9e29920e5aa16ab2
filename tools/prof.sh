#!/bin/bash
# Profile bench.py on the GPU box (run through gpurun):
#   tools/prof.sh <tag> [bench args...]
# pass 1: kernel trace + stats; pass 2/3: PMC FETCH_SIZE / WRITE_SIZE (separate passes, TCC slots).
# Raw output lands in gpurun_out/prof_<tag>/ (scratch); tools/prof_summary.py condenses it into profiles/.
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 "$@" > $out/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 "$@" > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 "$@" > $out/bench_write.log 2>&1
find $out -name "*.csv" | head -20
grep -h '^{"metric"' $out/bench_trace.log | cut -c1-300
