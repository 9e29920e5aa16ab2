#!/bin/bash
# r06 step 17: the whole suite at the round's last kernels (records of up to 32 k-mers, device parse), the profile set r06b, the raw-text stress,
# the emulated-rank stress on the longer records, N = 2 / 4 on a shared GPU
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s17; mkdir -p $o
timeout 3300 python -m pytest tests/ -x -q -m gpu --durations=6 2>&1 | tail -14 > $o/suite.log; cat $o/suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_set.sh r06b > $o/profile_set.log 2>&1; tail -3 $o/profile_set.log | cut -c1-300
timeout 1500 python3 tools/stress_raw.py 0 120 > $o/raw_text_seeds_0_119.log 2>&1; tail -1 $o/raw_text_seeds_0_119.log | cut -c1-200
timeout 1500 python3 tools/stress_multi_random.py 14000 150 > $o/emulated_ranks_seeds_14000_14149.log 2>&1; tail -1 $o/emulated_ranks_seeds_14000_14149.log
DSK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --check-parity --steps 3 --warmup 1 2>$o/bench_n2.err | grep '^{"metric"' > $o/bench_n2_shared_gpu_development.json; cut -c1-200 $o/bench_n2_shared_gpu_development.json
DSK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 4 --workload c2_10Mx150 --check-parity --steps 2 --warmup 1 2>$o/bench_n4.err | grep '^{"metric"' > $o/bench_n4_shared_gpu_development.json; cut -c1-200 $o/bench_n4_shared_gpu_development.json
