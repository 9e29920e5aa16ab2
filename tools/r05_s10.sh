#!/bin/bash
# round 5, session 10: count-table micro benchmark with the sum-checked variant (T3); N = 4 bench line in development mode on the 10 M-read workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s10
./tools/micro/count_tag > gpurun_out/r05_s10/count_tag.txt 2>&1; cat gpurun_out/r05_s10/count_tag.txt
DSK_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 4 --steps 2 --warmup 1 --check-parity --workload c2_10Mx150 > gpurun_out/r05_s10/bench_n4.json 2> gpurun_out/r05_s10/bench_n4.err
echo "N=4 rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r05_s10/bench_n4.json'))
for k in ('ms_per_step','rccl_ranks','exchange_alone','self_check','check_parity','sliced_steps'): print(k, d.get(k))"
