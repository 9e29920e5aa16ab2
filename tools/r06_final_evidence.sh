#!/bin/bash
# the round's last validation on a GPU box: every -m gpu test, smoke(), the default bench line (what the driver runs)
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_final; mkdir -p $o
timeout 3300 python -m pytest tests/ -x -q -m gpu --durations=6 2>&1 | tail -14 > $o/suite.log; cat $o/suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/smoke.log
( time python3 bench.py > $o/r06f_bench.json 2> $o/r06f_bench.err ) 2> $o/r06f_bench.time; tail -3 $o/r06f_bench.time
