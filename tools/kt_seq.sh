#!/bin/bash
# per-dispatch durations (in launch order) of the kernels matching a pattern: tools/kt_seq.sh <tag> "<pattern>" <script.py> [args]
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kt_$tag; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 "$@" > $out/run.log 2>&1
python3 - <<PY
import csv, re
rows=list(csv.DictReader(open("$out/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows:
    if re.search(r"$pat", r['Kernel_Name']):
        print(f"{r['Kernel_Name'][:60]:60s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r.get('Grid_Size_X','?')} wg {r.get('Workgroup_Size_X','?')} lds {r.get('LDS_Block_Size','?')} vgpr {r.get('VGPR_Count','?')}")
PY
