#!/bin/bash
# stress at HEAD: seeded random inputs against the CPU oracle -- two-word keys, one-word keys (rows in the reference's order: ascending
# inside every output partition; set STRESS_GLOBAL=1 for the global order), emulated multi-GPU ranks (the rebuilt sender).   usage: tools/stress.sh [tag]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r06}
out=gpurun_out/${tag}_stress; mkdir -p $out
if [ -z "$STRESS_GLOBAL" ]; then export STRESS_PARTITION=1; fi
timeout 1500 python3 tools/stress_random.py 9000 200 > $out/two_word_seeds_9000_9199.log 2>&1
tail -1 $out/two_word_seeds_9000_9199.log
STRESS_KS=15,21,27,31,32 timeout 1500 python3 tools/stress_random.py 10000 250 > $out/one_word_seeds_10000_10249.log 2>&1
tail -1 $out/one_word_seeds_10000_10249.log
timeout 1500 python3 tools/stress_multi_random.py 11000 200 > $out/emulated_ranks_seeds_11000_11199.log 2>&1
tail -1 $out/emulated_ranks_seeds_11000_11199.log
DSK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --check-parity --steps 3 --warmup 1 2>$out/bench_n2.err | grep '^{"metric"' > $out/bench_n2_shared_gpu_development.json; cut -c1-400 $out/bench_n2_shared_gpu_development.json
DSK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 4 --workload c2_10Mx150 --check-parity --steps 2 --warmup 1 2>$out/bench_n4.err | grep '^{"metric"' > $out/bench_n4_shared_gpu_development.json; cut -c1-400 $out/bench_n4_shared_gpu_development.json
