#!/bin/bash
# k = 63 profiles at the round's final kernels (top-word count table, 2560-key sub-partitions, rowsort2.h): kernel trace + PMC of the
# bench workload, 200 M reads on one GPU, the emulated rank of 8
tag=${1:-r04c}
bash tools/prof.sh ${tag}_k63pmc --kmer-size 63 --no-repeat-rich --steps 10 --warmup 3 > gpurun_out/prof_${tag}_k63pmc.log 2>&1
bash tools/kt_any.sh ${tag}_k63 bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --steps 10 --warmup 3 > gpurun_out/kt_${tag}_k63.log 2>&1
python3 tools/check_invariants.py c3_200Mx150 63 > gpurun_out/${tag}_c3_k63.txt 2>&1
python3 tools/mg_stage_times.py 8 63 0 c2_10Mx150 0 > gpurun_out/${tag}_mg_k63.txt 2>&1
python3 tools/mg_stage_times.py 8 63 0 c3_shard_25Mx150 0 > gpurun_out/${tag}_mg_shard_k63.txt 2>&1
tail -3 gpurun_out/${tag}_mg_k63.txt | cut -c1-300
