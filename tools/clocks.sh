#!/bin/bash
# clocks / power of the GPU while the bench runs (is the slow state of the scatters a clock state?):  tools/clocks.sh [steps]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
steps=${1:-400}
python3 bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-e2e --no-repeat-rich > gpurun_out/clocks_bench.log 2>&1 &
pid=$!
sleep 25
for i in $(seq 1 14); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|power\|Temperature (Sensor junction)\|hbm" | tr '\n' ';' | sed 's/GPU\[0\]//g; s/  */ /g'; echo
  sleep 0.7
done
wait $pid
tail -1 gpurun_out/clocks_bench.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
