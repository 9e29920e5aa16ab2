#!/bin/bash
# r06 step 16: the staging copy's thread count, A/B on one box: the dsk binary on the 3 GB FASTQ file, host parser and device parser
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r06_s16; mkdir -p $o
python3 - <<'PY'
import os, subprocess, sys, tempfile, time, re
import torch
sys.path.insert(0, os.getcwd())
import bench
from dsk_amd import synth
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp(prefix="dsk_e2e_")
gl, nr, rl = synth.workload("c2_10Mx150")
reads = synth.make_reads(synth.make_genome(gl, dev), nr, rl)
fq = os.path.join(tmp, "c2.fastq")
bench.write_fastq(reads, nr, rl, fq)
del reads
dsk = os.path.join(os.getcwd(), "dsk_amd", "host", "bin", "dsk")
for rnd in range(2):
    for extra in ((), ("-device-parse", "1")):
        for T in ("1", "2", "4", "8"):
            best = None
            for _ in range(3):
                if os.path.exists(os.path.join(tmp, "o.h5")): os.remove(os.path.join(tmp, "o.h5"))
                t0 = time.perf_counter()
                p = subprocess.run([dsk, "-file", fq, "-kmer-size", "31", "-abundance-min", "2", "-out", os.path.join(tmp, "o"), "-verbose", "1", *extra],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, DSKGPU_STAGE_THREADS=T))
                dt = time.perf_counter() - t0
                ing = float(re.search(r"ingest_s\s*:\s*([\d.]+)", p.stdout.decode()).group(1))
                if best is None or dt < best[0]: best = (round(dt, 3), ing)
            print("round", rnd, "device-parse" if extra else "host parser", "stage threads", T, "wall, ingest", best, flush=True)
PY
