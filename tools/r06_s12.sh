#!/bin/bash
# r06 step 12: records of up to 32 k-mers (the two 16-window halves of a packed word joined) and the device-side FASTA/FASTQ parser
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s12; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_raw_parse.py -x -q 2>&1 | tail -25 > $o/raw.log; cat $o/raw.log
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py -x -q -k "sender or super_kmer or multi_pass or group or slices or records or human_standin or multi_gpu or eight_ranks or exchange or receive" --durations=5 2>&1 | tail -25 > $o/rec.log; cat $o/rec.log
for a in "8 31 0 c2_10Mx150 0 partition" "8 63 0 c2_10Mx150 0 partition"; do python3 tools/mg_stage_times.py $a 2>&1 | tail -12 > "$o/mg_$(echo $a | cut -d' ' -f2).txt"; done
cat $o/mg_31.txt | cut -c1-400; cat $o/mg_63.txt | cut -c1-400
python3 tools/human_standin.py 2>&1 | tail -6 | cut -c1-600 > $o/standin.txt; cat $o/standin.txt
timeout 900 python3 tools/stress_multi_random.py 12000 120 > $o/emulated_ranks_12000.log 2>&1; tail -1 $o/emulated_ranks_12000.log
DSK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 4 --check-parity --steps 2 --warmup 1 2>$o/bench_n4.err | grep '^{"metric"' > $o/bench_n4.json; cut -c1-300 $o/bench_n4.json; tail -5 $o/bench_n4.err
