#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "human_standin" > $out/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_sel.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; tail -c 1500 $out/bench.json
DSK_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_n2.json 2> $out/bench_n2.err; echo "bench n2 rc $?"; tail -c 2500 $out/bench_n2.json
