#!/bin/bash
# round 5, session 1: the lifted caps (sender > 4.29 GB per rank, >= 2^32 rows) + neighbours
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s1
export DSKGPU_VERBOSE=
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_sort or sender or multi_pass_count_leaves or multi_gpu_path or group_count_in_one or step_in_slices or super_kmer_record or receive_side" 2>&1 | tail -15 > gpurun_out/r05_s1/small.log
cat gpurun_out/r05_s1/small.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "more_than_2_pow_32 or shards_above_4_gb or human_rank_shard" --durations=5 2>&1 | tail -40 > gpurun_out/r05_s1/big.log
cat gpurun_out/r05_s1/big.log
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "more_than_4_gb" --durations=3 2>&1 | tail -30 > gpurun_out/r05_s1/cli.log
cat gpurun_out/r05_s1/cli.log
