#!/bin/bash
# r05 step 13: first digit of the row sort on 11 bits (2048 buckets) against 10 -- sort tests with the variant, then A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s13; mkdir -p $out
DSKGPU_LIB=$PWD/dsk_amd/libdskgpu_a11.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort or skew or poly or golden or c2_small" 2>&1 | tail -5 > $out/parity_a11.log
cat $out/parity_a11.log
bash tools/ab_rep.sh "default dsk_amd/libdskgpu_a11.so" 3 > $out/ab.log 2>&1
cat $out/ab.log
bash tools/ab_rep.sh "default dsk_amd/libdskgpu_a11.so" 2 --kmer-size 63 > $out/ab63.log 2>&1
cat $out/ab63.log
