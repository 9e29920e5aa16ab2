#!/bin/bash
# the whole GPU suite, as the driver runs it, + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_suite
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=12 2>&1 | tail -45 > gpurun_out/r06_suite/suite.log
cat gpurun_out/r06_suite/suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
