#!/bin/bash
# round-4 GPU session 1: state of the tree on today's box + the stress of the sharded count with real processes
out=gpurun_out/r04a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
port=29600
for cfg in "2 31" "4 31" "2 63" "4 63"; do
  set -- $cfg; port=$((port+1))
  timeout 600 python -m torch.distributed.run --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/stress_multi.py $2 400000 50 > $out/stress_w$1_k$2.log 2>&1
  echo "stress world $1 k $2 rc $?"; grep -h "reference\|stress\|MISMATCH\|FAILED" $out/stress_w$1_k$2.log | head -8
done
port=$((port+1))
STRESS_RACE=1 timeout 600 python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $port tools/stress_multi.py 63 400000 3 > $out/stress_race_w4_k63.log 2>&1
echo "race demo rc $?"; grep -h "reference\|stress\|MISMATCH" $out/stress_race_w4_k63.log | head
