#!/bin/bash
# usage: r05_one.sh '<pytest -k expression>' [file]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_one
timeout 1500 python -m pytest ${2:-tests/} -x -q -m gpu -k "$1" --durations=5 2>&1 | tail -40 > gpurun_out/r05_one/one.log
cat gpurun_out/r05_one/one.log
