#!/bin/bash
# round 5, session 3: the N > 1 bench line with its self-checks (ranks sharing the one GPU: development mode), and the N = 1 line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s3
for n in 2 4; do
  DSK_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus $n --steps 2 --warmup 1 --check-parity > gpurun_out/r05_s3/bench_n$n.json 2> gpurun_out/r05_s3/bench_n$n.err
  echo "N=$n rc=$?"; tail -c 1500 gpurun_out/r05_s3/bench_n$n.err; head -c 3000 gpurun_out/r05_s3/bench_n$n.json; echo
done
timeout 900 python bench.py > gpurun_out/r05_s3/bench_n1.json 2> gpurun_out/r05_s3/bench_n1.err
echo "N=1 rc=$?"; head -c 2500 gpurun_out/r05_s3/bench_n1.json; echo
