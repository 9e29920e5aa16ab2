#!/bin/bash
# a long soak of the seeded random stress scripts against the CPU oracle (GPU box; ~45 min): new seed ranges on every call
#   usage: tools/soak.sh <tag> <seed base>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-soak}; b=${2:-20000}
o=gpurun_out/${tag}; mkdir -p $o
export STRESS_PARTITION=1
timeout 1500 python3 tools/stress_random.py $b 400 > $o/two_word_$b.log 2>&1; tail -n 1 $o/two_word_$b.log
STRESS_KS=15,21,27,31,32 timeout 1500 python3 tools/stress_random.py $((b+1000)) 400 > $o/one_word_$((b+1000)).log 2>&1; tail -n 1 $o/one_word_$((b+1000)).log
unset STRESS_PARTITION
STRESS_KS=65,72,80,96,97,101,127,128 timeout 1200 python3 tools/stress_random.py $((b+2000)) 80 > $o/four_word_$((b+2000)).log 2>&1; tail -n 1 $o/four_word_$((b+2000)).log
timeout 1500 python3 tools/stress_multi_random.py $((b+3000)) 300 > $o/emulated_ranks_$((b+3000)).log 2>&1; tail -n 1 $o/emulated_ranks_$((b+3000)).log
timeout 1500 python3 tools/stress_raw.py $((b+4000)) 400 > $o/raw_text_$((b+4000)).log 2>&1; tail -n 1 $o/raw_text_$((b+4000)).log
grep -L "stress ok" $o/*.log
