#!/bin/bash
# round 5, session 4: dskgpu_encode_reads -- the parity test, then configs[4]'s stand-in with and without the ASCII reads resident
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s4
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encode_reads or determinism" 2>&1 | tail -15
timeout 900 python tools/human_standin.py 600 31 2 2 > gpurun_out/r05_s4/standin_encoded.json 2> gpurun_out/r05_s4/standin_encoded.err; echo "rc=$?"; tail -c 600 gpurun_out/r05_s4/standin_encoded.err; cat gpurun_out/r05_s4/standin_encoded.json
timeout 900 python tools/human_standin.py 600 31 1 2 keep-ascii > gpurun_out/r05_s4/standin_ascii.json 2> gpurun_out/r05_s4/standin_ascii.err; echo "rc=$?"; cat gpurun_out/r05_s4/standin_ascii.json
