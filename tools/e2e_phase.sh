#!/bin/bash
# where a `dsk` run spends its wall clock outside execute() (DSK_PHASE_TIMES=1): tools/e2e_phase.sh [workload]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wl=${1:-ecoli50x}
python3 tools/e2e.py $wl 31 2>&1 | grep "dsk run" | tail -1
for i in 1 2 3; do
  s=$(date +%s.%N)
  DSK_PHASE_TIMES=1 dsk_amd/host/bin/dsk -file /tmp/e2e/$wl.fastq -kmer-size 31 -out /tmp/e2e/o2 -verbose 1 2>&1 | grep "\[dsk\]\|total_s\|ingest_s" | tr '\n' ' '
  e=$(date +%s.%N); echo " | process wall $(echo "$e - $s" | bc) s"
done
