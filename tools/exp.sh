#!/bin/bash
# One GPU-box session of an experiment: parity tests of the default build, then an A/B timing of several builds.
#   tools/exp.sh <tag> "<lib specs for tools/ab.sh>" [pytest -k expression]
tag=$1; libs=$2; kexpr=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
if [ "$kexpr" = none ]; then echo "pytest skipped" > $out/pytest.log
elif [ -n "$kexpr" ]; then timeout 2400 python3 -m pytest tests -x -q -m gpu -k "$kexpr" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
else timeout 2400 python3 -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; fi
tail -5 $out/pytest.log
bash tools/ab.sh "$libs" > $out/ab.log 2>&1
bash tools/ab.sh "$libs" >> $out/ab.log 2>&1
cat $out/ab.log
