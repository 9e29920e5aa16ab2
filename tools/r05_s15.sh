#!/bin/bash
# r05 step 15: width of the row sort's second digit (DSKGPU_RS_BBITS) at k = 31 and k = 63
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s15; mkdir -p $out
bash tools/ab_rep.sh "default DSKGPU_RS_BBITS=9:default DSKGPU_RS_BBITS=10:default" 2 > $out/ab31.log 2>&1
cat $out/ab31.log
bash tools/ab_rep.sh "default DSKGPU_RS_BBITS=9:default DSKGPU_RS_BBITS=10:default" 2 --kmer-size 63 > $out/ab63.log 2>&1
cat $out/ab63.log
