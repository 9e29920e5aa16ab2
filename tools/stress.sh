#!/bin/bash
# round-5 stress at HEAD: seeded random inputs against the CPU oracle -- two-word keys, one-word keys, emulated multi-GPU ranks
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_stress; mkdir -p $out
timeout 1500 python3 tools/stress_random.py 6000 250 > $out/two_word_seeds_6000_6249.log 2>&1
tail -1 $out/two_word_seeds_6000_6249.log
STRESS_KS=15,21,27,31,32 timeout 1500 python3 tools/stress_random.py 7000 250 > $out/one_word_seeds_7000_7249.log 2>&1
tail -1 $out/one_word_seeds_7000_7249.log
timeout 1500 python3 tools/stress_multi_random.py 8000 200 > $out/emulated_ranks_seeds_8000_8199.log 2>&1
tail -1 $out/emulated_ranks_seeds_8000_8199.log
