#!/bin/bash
# r05 step 18: valid k-mer windows counted by the encode kernel (k <= 32) -- the whole GPU suite, then A/B against the build before
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s18; mkdir -p $out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > $out/suite.log
cat $out/suite.log
bash tools/ab_rep.sh "dsk_amd/libdskgpu_prev.so default" 3 > $out/ab.log 2>&1
cat $out/ab.log
