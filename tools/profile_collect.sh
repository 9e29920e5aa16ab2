#!/bin/bash
# copy what tools/profile_set.sh <tag> left under gpurun_out/ (scratch) into profiles/ (tracked): run HERE after the gpurun call
tag=${1:?tag}
python3 tools/prof_summary.py $tag > /dev/null
python3 tools/prof_summary.py ${tag}_k63pmc > /dev/null
stats() { python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:40]: w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
}
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
stats gpurun_out/kt_${tag}_k63/t_kernel_stats.csv profiles/${tag}_k63_kernel_stats.csv
grep -h '^{"metric"' gpurun_out/kt_${tag}_k63/run.log | tail -1 > profiles/${tag}_k63_bench_c2_k63.json
stats gpurun_out/kt_${tag}_human/t_kernel_stats.csv profiles/${tag}_human_standin_kernel_stats.csv
grep -h '^{"workload"' gpurun_out/kt_${tag}_human/run.log | tail -1 > profiles/${tag}_human_standin.json
stats gpurun_out/kt_${tag}_c3/t_kernel_stats.csv profiles/${tag}_c3_200Mx150_kernel_stats.csv
grep -v '^[WE]2026' gpurun_out/kt_${tag}_c3/run.log | grep -v amdgpu.ids > profiles/${tag}_c3_200Mx150.txt
stats gpurun_out/kt_${tag}_mg/t_kernel_stats.csv profiles/${tag}_multigpu_rank_kernel_stats.csv
{ for f in gpurun_out/kt_${tag}_mg/run.log gpurun_out/${tag}_mg_sliced.txt gpurun_out/${tag}_mg_shard.txt gpurun_out/${tag}_mg_k63.txt; do echo "# $f"; grep -v '^[WE]2026' $f | grep -v amdgpu.ids; done; } > profiles/${tag}_multigpu_rank_stage_times.txt
ls -la profiles | grep $tag
# k = 63 refresh (tools/profile_set.sh <tag2>): profiles/<tag2>_k63*
if [ -n "$2" ]; then
  t2=$2
  python3 tools/prof_summary.py ${t2}_k63pmc > /dev/null
  stats gpurun_out/kt_${t2}_k63/t_kernel_stats.csv profiles/${t2}_k63_kernel_stats.csv
  grep -h '^{"metric"' gpurun_out/kt_${t2}_k63/run.log | tail -1 > profiles/${t2}_k63_bench_c2_k63.json
  { for f in gpurun_out/${t2}_c3_k63.txt gpurun_out/${t2}_mg_k63.txt gpurun_out/${t2}_mg_shard_k63.txt; do echo "# $f"; grep -v '^[WE]2026' $f | grep -v amdgpu.ids; done; } > profiles/${t2}_k63_large_inputs_and_ranks.txt
  ls -la profiles | grep ${t2}_k63
fi
