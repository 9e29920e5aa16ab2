#!/bin/bash
# r05 step 17: minimizer order hash of the super-k-mer sender: one multiply + fold (default) against the two-multiply finaliser (libdskgpu_fmix.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_s17; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "super_kmer or repartition or group or multi or sender or minimizer or emulated or rank or table or human or pass" 2>&1 | tail -6 > $out/parity.log
cat $out/parity.log
for r in 1 2; do
  for lib in dsk_amd/libdskgpu_fmix.so dsk_amd/libdskgpu.so; do
    echo "== $lib"
    DSKGPU_LIB=$PWD/$lib python3 tools/mg_stage_times.py 8 31 0 c2_10Mx150 0 2>&1 | grep "world=\|stages" | cut -c1-330
  done
done > $out/mg.log 2>&1
cat $out/mg.log
for lib in dsk_amd/libdskgpu_fmix.so dsk_amd/libdskgpu.so; do
  echo "== $lib"
  DSKGPU_LIB=$PWD/$lib python3 tools/human_standin.py 600 31 1 2>&1 | grep '^{"workload"' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('seconds','n_passes','n_read_sweeps','hbm_used_gb','stage_ms','kmer_occurrences_per_s')})"
done > $out/human.log 2>&1
cat $out/human.log
