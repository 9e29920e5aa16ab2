#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2b; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or synthetic or randomized or histogram_free or poly or abundance_window or determinism or full_size or group_count or rccl" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout 600 python3 -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "nb_gpus or auto" > $out/pytest_cli.log 2>&1; echo "pytest rc=$?" >> $out/pytest_cli.log
tail -15 $out/pytest_cli.log
tools/ab.sh "DSKGPU_NO_PACKED_COUNT=1:default default dsk_amd/variants/c_1024_4_8.so dsk_amd/variants/c_1024_2_8.so dsk_amd/variants/c_512_4_6.so dsk_amd/variants/c_512_6_4.so dsk_amd/variants/c_512_4_8t.so dsk_amd/variants/c_256_8_8.so" 2>&1 | tee $out/ab.log
timeout 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 > $out/bench_gpus2.log 2>&1; echo "bench --gpus 2 rc=$?"; tail -3 $out/bench_gpus2.log | cut -c1-400
