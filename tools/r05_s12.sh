#!/bin/bash
# r05 step 12: setup kernel also zeroes the level-2 fill counts -- parity subset, A/B, kernel trace of a step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s12; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -8 > $out/parity.log
cat $out/parity.log
bash tools/ab_rep.sh "dsk_amd/libdskgpu_base.so default" 3 > $out/ab.log 2>&1
cat $out/ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --no-cpu-baseline --no-e2e --no-human-standin --no-place-compare --no-k63 --steps 6 --warmup 2 > $out/bench_trace.log 2>&1
python3 - <<'PY' > $out/timeline.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/r05_s12/trace/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('void k_encode')]
a,b=idx[-2],idx[-1]
prev=None; tg=0
for r in rows[a:b]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap=(s-prev)/1000 if prev else 0; tg+=gap
    print(f"{gap:8.1f} us gap | {(e-s)/1000:9.1f} us | {r['Kernel_Name'][:80]}")
    prev=e
print('dispatches',b-a,'gaps us',round(tg,1))
PY
cat $out/timeline.txt
rm -rf $out/trace
