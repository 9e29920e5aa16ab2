#!/bin/bash
# k = 63 evidence (configs[3]: two-word keys): bench line + kernel trace of the bench workload and of one GPU's 25 M-read share
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/k63; mkdir -p $out
python3 bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-repeat-rich --steps 8 --warmup 2 > $out/bench_c2_k63.json 2> $out/err.log
python3 bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-repeat-rich --steps 4 --warmup 1 --workload c3_shard_25Mx150 > $out/bench_shard_k63.json 2>> $out/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-repeat-rich --steps 6 --warmup 2 > $out/bench_trace.log 2>&1
python3 tools/check_invariants.py c3_200Mx150 63 2>&1 | grep -v "^\[(" | cut -c1-600 | tail -2
for f in $out/bench_c2_k63.json $out/bench_shard_k63.json; do python3 -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['config']['workload'][:40], round(d['ms_per_step'],3), int(d['n_kmers']), {k:round(v,3) for k,v in d['stage_ms'].items()}, d['roofline']['kernel'], d['roofline']['frac'])"; done
