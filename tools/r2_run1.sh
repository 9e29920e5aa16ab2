#!/bin/bash
# round-2 GPU session 1: parity of the packed count kernel, A/B of its variants, SQ counters of the three big kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r2a; mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or synthetic or randomized or histogram_free or poly or abundance_window or determinism or full_size" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
tools/ab.sh "DSKGPU_NO_PACKED_COUNT=1:default default dsk_amd/variants/cp_512_6_4.so dsk_amd/variants/cp_1024_4_8.so dsk_amd/variants/cp_256_12_4.so" 2>&1 | tee $out/ab.log
tools/pmc.sh r2a_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" 2>&1 | tail -12 | tee $out/pmc1.log
tools/pmc.sh r2a_sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" 2>&1 | tail -12 | tee $out/pmc2.log
