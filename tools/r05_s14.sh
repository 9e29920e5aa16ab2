#!/bin/bash
# r05 step 14: two-word row sort, first digit on 10 / 11 (default) / 12 bits -- parity of the default, then A/B at k = 63
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s14; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -5 > $out/parity.log
cat $out/parity.log
bash tools/ab_rep.sh "dsk_amd/libdskgpu_r2a10.so default dsk_amd/libdskgpu_r2a12.so" 3 --kmer-size 63 > $out/ab63.log 2>&1
cat $out/ab63.log
