#!/bin/bash
# Full GPU check as the driver does it: all -m gpu tests, smoke(), the default bench line.   usage: tools/gpu_full.sh <tag>
tag=${1:-full}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
timeout 2400 python3 -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python3 bench.py ) > $out/bench.log 2>&1; tail -4 $out/bench.log | cut -c1-3000
