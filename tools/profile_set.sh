#!/bin/bash
# the profile set of a round (GPU box): kernel trace + PMC of the bench workload, kernel trace at k = 63, of the human stand-in, and the
# emulated rank of an 8-GPU job.  Raw output under gpurun_out/; tools/prof_summary.py + the copies below condense it into profiles/.
tag=${1:?tag, e.g. r06a}
bash tools/prof.sh $tag --no-repeat-rich --steps 20 --warmup 5 > gpurun_out/prof_$tag.log 2>&1
python3 bench.py --no-human-standin > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
bash tools/prof.sh ${tag}_k63pmc --kmer-size 63 --no-repeat-rich --steps 10 --warmup 3 > gpurun_out/prof_${tag}_k63pmc.log 2>&1
bash tools/kt_any.sh ${tag}_k63 bench.py --kmer-size 63 --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-repeat-rich --steps 10 --warmup 3 > gpurun_out/kt_${tag}_k63.log 2>&1
bash tools/kt_any.sh ${tag}_human tools/human_standin.py 600 31 1 > gpurun_out/kt_${tag}_human.log 2>&1
bash tools/kt_any.sh ${tag}_c3 tools/check_invariants.py c3_200Mx150 31 > gpurun_out/kt_${tag}_c3.log 2>&1
bash tools/kt_any.sh ${tag}_mg tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 partition > gpurun_out/kt_${tag}_mg.log 2>&1
python3 tools/mg_stage_times.py 8 31 0 c2_10Mx150 4 partition > gpurun_out/${tag}_mg_sliced.txt 2>&1
python3 tools/mg_stage_times.py 8 31 0 c3_shard_25Mx150 0 partition > gpurun_out/${tag}_mg_shard.txt 2>&1
python3 tools/mg_stage_times.py 8 63 0 c2_10Mx150 0 partition > gpurun_out/${tag}_mg_k63.txt 2>&1
tail -3 gpurun_out/kt_${tag}_mg.log gpurun_out/${tag}_mg_shard.txt | cut -c1-400
