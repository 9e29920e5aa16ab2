#!/bin/bash
# round 5, session 8: level 1 without spills (dump-zone base opaque per chunk) against the build before it -- k = 31 and k = 63, repeat-rich twin
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s8
bash tools/ab_rep.sh "dsk_amd/libdskgpu_abl.so default" 3 --no-human-standin --no-k63 > gpurun_out/r05_s8/ab_k31.txt 2>&1
cat gpurun_out/r05_s8/ab_k31.txt
bash tools/ab_rep.sh "dsk_amd/libdskgpu_abl.so default" 2 --no-human-standin --no-k63 --no-repeat-rich --kmer-size 63 > gpurun_out/r05_s8/ab_k63.txt 2>&1
cat gpurun_out/r05_s8/ab_k63.txt
