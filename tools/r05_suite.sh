#!/bin/bash
# the whole GPU suite, as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_suite
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/r05_suite/suite.log
cat gpurun_out/r05_suite/suite.log
