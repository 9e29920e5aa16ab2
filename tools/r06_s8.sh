#!/bin/bash
# r06 step 8: partition order in multi-pass jobs (tests + stand-in both ways) and the parallel gzip's phases on this host
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s8; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "partition_order or multi_pass or human_standin or full_size_multi" 2>&1 | tail -6 | tee $o/tests.txt
python3 tools/human_standin.py 600 31 1 2 - partition > $o/hs_part.txt 2>&1; tail -1 $o/hs_part.txt | cut -c1-1300
python3 tools/human_standin.py 600 31 1 2 - global > $o/hs_glob.txt 2>&1; tail -1 $o/hs_glob.txt | cut -c1-1300
# gzip e2e with the phase trace
python3 - <<'PY'
import os, subprocess, sys, time, gzip, shutil
sys.path.insert(0, ".")
import torch
from dsk_amd import synth
tmp = "/tmp/r06_s8"; os.makedirs(tmp, exist_ok=True)
dev = torch.device("cuda:0")
gl, nr, rl = synth.workload("ecoli50x")
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl).cpu().numpy().reshape(-1, rl + 1)
q = b"I" * rl
with open(tmp + "/e.fq", "wb") as f:
    for i in range(nr):
        f.write(b"@r%d\n" % i + reads[i, :rl].tobytes() + b"\n+\n" + q + b"\n")
subprocess.check_call("gzip -kf -6 %s/e.fq" % tmp, shell=True)
print("sizes", os.path.getsize(tmp + "/e.fq"), os.path.getsize(tmp + "/e.fq.gz"))
dsk = "dsk_amd/host/bin/dsk"
for env in ({"DSK_PGZIP_TRACE": "1"}, {"DSK_PGZIP_TRACE": "1"}, {"DSK_NO_PGZIP": "1"}):
    t0 = time.perf_counter()
    r = subprocess.run([dsk, "-file", tmp + "/e.fq.gz", "-kmer-size", "31", "-out", tmp + "/e", "-verbose", "1"], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print(env, "wall %.3f s rc %d" % (dt, r.returncode)); print(r.stderr.decode()[-900:]); print([l for l in r.stdout.decode().splitlines() if "time" in l or "_s" in l][:12])
shutil.rmtree(tmp)
PY
